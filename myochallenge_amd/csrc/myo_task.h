// myo_task.h — Baoding task layer on the device: goal schedule, action map, observation,
// reward, termination, TimeLimit, SubprocVecEnv-style auto-reset with the reference's
// curriculum / RSI / domain-randomisation reset logic.
//
// Restates (per environment, wave-uniform control flow):
//   BaodingEnvV1.step + BaseV0.step            [MyoSuite 1.2.3, SURVEY.md §3.3, Appendix B.8]
//   CustomBaodingEnv.get_reward_dict           /root/reference/src/envs/baoding.py:24-94 (P2: 403-467)
//   CustomBaodingEnv.reset                     /root/reference/src/envs/baoding.py:146-208
//   CustomBaodingP2Env.reset                   /root/reference/src/envs/baoding.py:494-647
//   _add_noise_to_{palm,finger}_positions      /root/reference/src/envs/baoding.py:96-144,469-492
//   TimeLimit(200) + SubprocVecEnv auto-reset  /root/reference/src/envs/__init__.py:15,61; SURVEY C.6
// Die reorient (kind MYO_TASK_REORIENT), same structure:
//   CustomReorientEnv.get_reward_dict          /root/reference/src/envs/reorient.py:12-56
//   CustomReorientEnv.reset / set_orientation  /root/reference/src/envs/reorient.py:124-205
//   CustomReorientEnv.step (shaping state)     /root/reference/src/envs/reorient.py:207-212
//   ReorientEnvV0.get_obs_dict, euler2quat / mat2euler  [MyoSuite 1.2.3, 3P-RECALL]
// Random numbers: the reference mixes gym's np_random, the global np.random and random.choice
// (SURVEY §7.4-6); seed-for-seed parity is not meaningful, so the device draws from
// Philox4x32-10 keyed by (seed, env, episode) in the reference's draw ORDER.
#pragma once
#include "myo_physics.h"

#define MYO_PI 3.14159265358979323846

struct Philox {
  unsigned int key0, key1, c0, c1, c2, idx;
};
DEV void philox_round(unsigned int* c, unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned int h0 = (unsigned int)(p0 >> 32), l0 = (unsigned int)p0, h1 = (unsigned int)(p1 >> 32), l1 = (unsigned int)p1;
  const unsigned int n0 = h1 ^ c[1] ^ k0, n1 = l1, n2 = h0 ^ c[3] ^ k1, n3 = l0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
DEV double philox_uniform(Philox& g) {  // (0,1), 53 bits
  unsigned int c[4] = {g.c0, g.c1, g.c2, g.idx};
  unsigned int k0 = g.key0, k1 = g.key1;
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  g.idx++;
  const unsigned long long bits = ((((unsigned long long)c[0]) << 32) | (unsigned long long)c[1]) >> 11;
  return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}
DEV double rng_range(Philox& g, double lo, double hi) { return lo + (hi - lo) * philox_uniform(g); }
DEV double rng_normal(Philox& g) {
  const double u1 = philox_uniform(g), u2 = philox_uniform(g);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * MYO_PI * u2);
}
DEV double rng_gamma(Philox& g, double a) {  // Marsaglia-Tsang
  double boost = 1.0;
  if (a < 1.0) { boost = pow(philox_uniform(g), 1.0 / a); a += 1.0; }
  const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (int it = 0; it < 64; ++it) {
    const double x = rng_normal(g);
    double v = 1.0 + c * x;
    if (v <= 0) continue;
    v = v * v * v;
    const double u = philox_uniform(g);
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
  }
  return boost * d;
}
DEV double rng_beta(Philox& g, double a, double b) {
  const double x = rng_gamma(g, a), y = rng_gamma(g, b);
  return x / (x + y);
}

// ---- observation (layout pinned by tests/golden/reset_obs_golden.npy) and reward
template <typename T, int NC>
DEVFN void baoding_obs_reward(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  const int nh = K.n_hand;
  const HP dt = (HP)K.frame_skip * M.h_timestep;
  // positions, errors and velocities from the HP state and poses (world coordinates), rounded once to T
  PHASE {
    const int i = lane;
    if (i < nh) S_OBS(s)[i] = (T)s.qpos[i];
    if (i < 3) {
      HP p1[3], p2[3], t1[3], t2[3];
      body_point_hp(s, M.site_bodyid[K.obj1_sid], M.h_site_pos + 3 * K.obj1_sid, p1);
      body_point_hp(s, M.site_bodyid[K.obj2_sid], M.h_site_pos + 3 * K.obj2_sid, p2);
      const HP l1[3] = {s.target_xy[0], s.target_xy[1], M.h_site_pos[3 * K.target1_sid + 2]};
      const HP l2[3] = {s.target_xy[2], s.target_xy[3], M.h_site_pos[3 * K.target2_sid + 2]};
      body_point_hp(s, M.site_bodyid[K.target1_sid], l1, t1);
      body_point_hp(s, M.site_bodyid[K.target2_sid], l2, t2);
      // (lane-indexed reads of the local arrays would put them in private memory: select instead)
      const HP p1i = i == 0 ? p1[0] : (i == 1 ? p1[1] : p1[2]), p2i = i == 0 ? p2[0] : (i == 1 ? p2[1] : p2[2]);
      const HP t1i = i == 0 ? t1[0] : (i == 1 ? t1[1] : t1[2]), t2i = i == 0 ? t2[0] : (i == 1 ? t2[1] : t2[2]);
      S_OBS(s)[nh + i] = (T)p1i;
      S_OBS(s)[nh + 3 + i] = (T)(s.qvel[M.nv - 12 + i] * dt);
      S_OBS(s)[nh + 6 + i] = (T)p2i;
      S_OBS(s)[nh + 9 + i] = (T)(s.qvel[M.nv - 6 + i] * dt);
      S_OBS(s)[nh + 12 + i] = (T)t1i;
      S_OBS(s)[nh + 15 + i] = (T)t2i;
      S_OBS(s)[nh + 18 + i] = (T)(t1i - p1i);
      S_OBS(s)[nh + 21 + i] = (T)(t2i - p2i);
      s.target_w[i] = t1i; s.target_w[3 + i] = t2i;
    }
    if (i < M.na) S_OBS(s)[nh + 24 + i] = (T)S_ACT(M, s)[i];
  }
  SYNC();
  WAVE_SUM_N(T, asq, M.na, i, ((T)S_ACT(M, s)[i] * (T)S_ACT(M, s)[i]));
  PHASE {
    if (lane == 0) {
      const T* e1 = S_OBS(s) + nh + 18; const T* e2 = S_OBS(s) + nh + 21;
      const T d1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), d2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
      const T am = M.na ? sqrt(asq) / (T)M.na : (T)0;
      const int fall = (S_OBS(s)[nh + 2] < (T)K.drop_th) || (S_OBS(s)[nh + 8] < (T)K.drop_th);
      T c[7];
      c[0] = -d1; c[1] = -d2; c[2] = -am; c[3] = fall ? (T)0 : (T)1; c[4] = -(d1 + d2);
      c[5] = ((d1 < (T)K.proximity_th) && (d2 < (T)K.proximity_th) && !fall) ? (T)1 : (T)0;
      c[6] = fall ? (T)1 : (T)0;
      T dense = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) { dense += (T)K.weights[k] * c[k]; S_RWD(s)[k] = c[k]; }
      S_RWD(s)[7] = dense;
    }
  }
  SYNC_G();      // (Scratch::SPILL keeps the reward terms in global memory: every lane reads them)
}

// ---- die reorient: rotations in the mujoco-py rotations.py convention MyoSuite's quat_math copies [3P-RECALL]
DEV void ro_euler2quat(const HP* e, HP* q) {
  const HP ai = e[2] / 2, aj = -e[1] / 2, ak = e[0] / 2;
  const HP si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
  const HP cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = cj * cc + sj * ss; q[1] = cj * cs - sj * sc; q[2] = -(cj * ss + sj * cc); q[3] = cj * sc - sj * cs;
}
DEV void ro_quat2euler(const HP* qin, HP* e) {    // mat2euler(quat2mat(q)), q normalised first
  const HP n = sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
  const HP w = qin[0] / n, x = qin[1] / n, y = qin[2] / n, z = qin[3] / n;
  const HP m00 = 1 - 2 * (y * y + z * z), m01 = 2 * (x * y - w * z), m02 = 2 * (x * z + w * y);
  const HP m10 = 2 * (x * y + w * z), m11 = 1 - 2 * (x * x + z * z), m12 = 2 * (y * z - w * x);
  const HP m22 = 1 - 2 * (x * x + y * y);
  const HP cy = sqrt(m22 * m22 + m12 * m12);
  const int cond = cy > 8.881784197001252e-16;   // _EPS4 = 4 * float64 eps
  e[2] = cond ? -atan2(m01, m00) : -atan2(-m10, m11);
  e[1] = -atan2(-m02, cy);
  e[0] = cond ? -atan2(m12, m22) : (HP)0;
}

// observation = hand_qpos, hand_qvel * dt, obj_pos, goal_pos, pos_err, obj_rot, goal_rot, rot_err, act; reward dictionary of
// reorient.py:12-56 with the shaping terms taken against the distances of the previous step (or of the reset), which
// are replaced by this step's at the end (reorient.py:178-179,207-210).  comps = pos_dist, rot_dist, act_reg, alive,
// sparse, solved, done, dense (the two *_diff terms enter dense only).
template <typename T, int NC>
DEVFN void reorient_obs_reward(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  const int nh = K.n_hand;
  const HP dt = (HP)K.frame_skip * M.h_timestep;
  T* o = S_OBS(s);
  PHASE {
    const int i = lane;
    if (i < nh) { o[i] = (T)s.qpos[i]; o[nh + i] = (T)(s.qvel[i] * dt); }
    if (i < M.na) o[2 * nh + 18 + i] = (T)S_ACT(M, s)[i];
  }
  SYNC();
  WAVE_SUM_N(T, asq, M.na, i, ((T)S_ACT(M, s)[i] * (T)S_ACT(M, s)[i]));
  PHASE {
    if (lane == 0) {
      HP op[3], gp[3], er[3], oe[3], ge[3], t[3];
      body_point_hp(s, K.ro_obj_bid, M.h_site_pos + 3 * K.obj1_sid, op);
      {  // the goal body hangs off the world with the episode's pose: site target_o = goal_pos + R(goal_quat) * site_pos
        const HP* q = s.goal_quat; const HP* l = M.h_site_pos + 3 * K.target1_sid;
        const HP n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const HP w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        t[0] = (1 - 2 * (y * y + z * z)) * l[0] + 2 * (x * y - w * z) * l[1] + 2 * (x * z + w * y) * l[2];
        t[1] = 2 * (x * y + w * z) * l[0] + (1 - 2 * (x * x + z * z)) * l[1] + 2 * (y * z - w * x) * l[2];
        t[2] = 2 * (x * z - w * y) * l[0] + 2 * (y * z + w * x) * l[1] + (1 - 2 * (x * x + y * y)) * l[2];
        for (int k = 0; k < 3; ++k) gp[k] = s.goal_pos[k] + t[k];
      }
      ro_quat2euler(S_XQUAT(s) + 4 * K.ro_obj_bid, oe);
      ro_quat2euler(s.goal_quat, ge);
      HP pd = 0, rd = 0;
      for (int k = 0; k < 3; ++k) {
        er[k] = gp[k] - op[k] - K.ro_goal_obj_offset[k];
        const HP re = ge[k] - oe[k];
        o[2 * nh + k] = (T)op[k]; o[2 * nh + 3 + k] = (T)gp[k]; o[2 * nh + 6 + k] = (T)er[k];
        o[2 * nh + 9 + k] = (T)oe[k]; o[2 * nh + 12 + k] = (T)ge[k]; o[2 * nh + 15 + k] = (T)re;
        pd += er[k] * er[k]; rd += re * re;
      }
      pd = sqrt(pd); rd = sqrt(rd);
      const HP am = M.na ? sqrt((HP)asq) / (HP)M.na : (HP)0;
      const int drop = pd > K.drop_th;
      HP c[9];
      c[0] = -pd; c[1] = -rd; c[2] = s.pos_dist - pd; c[3] = s.rot_dist - rd; c[4] = drop ? 0 : 1; c[5] = -am;
      c[6] = -rd - 10.0 * pd; c[7] = (pd < K.ro_pos_th && rd < K.ro_rot_th && !drop) ? 1 : 0; c[8] = drop ? 1 : 0;
      HP dense = 0;
      for (int k = 0; k < 9; ++k) dense += K.ro_weights[k] * c[k];
      S_RWD(s)[0] = (T)c[0]; S_RWD(s)[1] = (T)c[1]; S_RWD(s)[2] = (T)c[5]; S_RWD(s)[3] = (T)c[4]; S_RWD(s)[4] = (T)c[6];
      S_RWD(s)[5] = (T)c[7]; S_RWD(s)[6] = (T)c[8]; S_RWD(s)[7] = (T)dense;
      s.pos_dist = pd; s.rot_dist = rd;
    }
  }
  SYNC_G();      // (Scratch::SPILL keeps the reward terms in global memory: every lane reads them)
}

template <typename T, int NC>
DEV void task_obs_reward(const DevModel<T>& M, const TaskDev& K, Scratch<T, NC>& s) {
  if (K.kind == MYO_TASK_REORIENT_K) reorient_obs_reward(M, K, s); else baoding_obs_reward(M, K, s);
}
DEV int task_nobs(const TaskDev& K, int na) { return K.kind == MYO_TASK_REORIENT_K ? 2 * K.n_hand + 18 + na : K.n_hand + 24 + na; }

// ---- env.step(a) without the VecEnv bookkeeping
// An env step may be run in parts by several workgroups (the launch's makespan, see k_step): substeps [k_lo, k_hi) of the
// frame_skip; the part that starts at 0 also moves the targets, every part recomputes ctrl (a pure function of the action), the
// part that ends at frame_skip also makes observation and reward (k_hi < 0 = frame_skip).  Everything a substep hands to the
// next one is in the env record (load_env / store_env), so the split is bit-exact.
template <typename T, int RKM = -1, int NC>
DEV void task_step_core(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, const float* action /* may be null = zeros */,
                        int k_lo = 0, int k_hi = -1) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  PHASE {
    if (lane == 0 && K.kind != MYO_TASK_REORIENT_K && k_lo == 0) {
      if (s.which_task != 0) {
        const double dt = (double)K.frame_skip * M.h_timestep;
        const double sign = s.which_task == 1 ? -1.0 : 1.0;
        const double ang = sign * 2.0 * MYO_PI * ((double)s.counter * dt / (double)s.time_period);
        const double a1 = ang + s.start_angle[0], a2 = ang + s.start_angle[1];
        s.target_xy[0] = s.x_radius * cos(a1) + K.center_pos[0];
        s.target_xy[1] = s.y_radius * sin(a1) + K.center_pos[1];
        s.target_xy[2] = s.x_radius * cos(a2) + K.center_pos[0];
        s.target_xy[3] = s.y_radius * sin(a2) + K.center_pos[1];
      }
      s.counter++;
    }
    const int i = lane;
    if (i < M.nu) {  // BaseV0.step: clip, then float32 sigmoid(5(a-0.5)) for muscles (normalize_act)
      float a = action ? action[i] : 0.0f;
      a = a < -1.f ? -1.f : (a > 1.f ? 1.f : a);
      // every operation but exp is an IEEE float32 operation; exp is the correctly rounded float32 exponential
      // (a device expf is 1 ulp off glibc's / numpy's on some arguments: a 6e-8 difference in ctrl, 1e-8 in the
      // trajectory after 20 env steps)
      const float c = 1.0f / (1.0f + (float)exp((double)(-5.0f * (a - 0.5f))));
      ctrl_set(s, i, (T)c);
    }
  }
  SYNC();
  const int k_end = (k_hi < 0 || k_hi > K.frame_skip) ? K.frame_skip : k_hi;
  for (int k = k_lo; k < k_end; ++k) mj_step<T, RKM>(M, K, s);
  if (k_end < K.frame_skip) return;
  check_state(M, s, 0);            // a non-finite value produced by the LAST advance must not leave through obs / reward
  kinematics(M, s);
  task_obs_reward(M, K, s);
}

template <typename T, int NC>
DEVFN void set_init_state(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int keep_dynamics) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  // robot.reset(init_qpos, init_qvel): init_qpos[:-14]=0, init_qpos[0]=-1.57 (baoding.py:281-283)
  WAVE_FN
  PHASE {
    for (int i = lane; i < M.nq; i += 64) s.qpos[i] = (i < K.n_hand) ? (i == 0 ? (HP)K.init_qpos0 : (HP)0) : (HP)M.h_qpos0[i];
    for (int i = lane; i < M.nv; i += 64) { s.qvel[i] = 0; if (!keep_dynamics) warm_set(s, i, (T)0); }
    if (!keep_dynamics) {
      for (int i = lane; i < M.na; i += 64) act_set(M, s, i, (HP)0);
      for (int i = lane; i < M.nu; i += 64) ctrl_set(s, i, (T)0);
      if (lane == 0) { s.time = 0; s.bad = 0; }
    }
  }
  SYNC();
}

// ---- reset(): returns with the post-reset state in scratch and the reset observation in S_OBS(s)
template <typename T, int NC>
DEVFN void baoding_reset(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int env) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  Philox g;
  g.key0 = (unsigned int)K.seed; g.key1 = (unsigned int)(K.seed >> 32);
  g.c0 = (unsigned int)env; g.c1 = (unsigned int)s.episode; g.c2 = 0x42414f44u; g.idx = 0;
  int do_rsi = 0;
  // every lane runs the (cheap, scalar) sampling redundantly with identical results; lane 0 stores
  double start1 = (double)s.start_angle[0], start2 = (double)s.start_angle[1];
  int which = s.which_task;
  double xr, yr, period, mass[2], fric[6], size[2];
  mass[0] = (double)s.ball_mass[0]; mass[1] = (double)s.ball_mass[1];
  size[0] = (double)s.ball_size[0]; size[1] = (double)s.ball_size[1];
  for (int k = 0; k < 6; ++k) fric[k] = (double)s.ball_fric[k];
  if (K.kind == 1) {
    // P1 (baoding.py:146-208)
    if (K.task_choice == 3) { int c = (int)(philox_uniform(g) * 3.0); which = c > 2 ? 2 : c; }
    else which = (K.task_choice == 1) ? 1 : 2;
    const double phase = K.enable_rsi ? rng_range(g, -MYO_PI, MYO_PI) : 0.0;
    start1 = 3.0 * MYO_PI / 4.0 + phase; start2 = -MYO_PI / 4.0 + phase;
    xr = rng_range(g, K.goal_xrange[0], K.goal_xrange[1]);
    yr = rng_range(g, K.goal_yrange[0], K.goal_yrange[1]);
    period = rng_range(g, K.goal_time_period[0], K.goal_time_period[1]);
    if (K.enable_rsi) do_rsi = philox_uniform(g) < K.rsi_probability;
  } else {
    // P2 (baoding.py:494-647)
    if (K.task_choice == 3) {
      int c = (int)(philox_uniform(g) * 3.0);
      which = c > 2 ? 2 : c;
      if (philox_uniform(g) < K.overlap_probability) start1 = 3.0 * MYO_PI / 4.0;
      else if (K.limit_init_angle_on) {
        double phase = rng_range(g, -K.limit_init_angle, K.limit_init_angle);
        if (K.beta_init_angle_on) phase = rng_beta(g, K.beta_init_angle[0], K.beta_init_angle[1]) * 2.0 * MYO_PI - MYO_PI;
        start1 = 3.0 * MYO_PI / 4.0 + phase;
      } else start1 = rng_range(g, 0.0, 2.0 * MYO_PI);
      start2 = start1 - MYO_PI;
    }
    xr = rng_range(g, K.goal_xrange[0], K.goal_xrange[1]);
    yr = rng_range(g, K.goal_yrange[0], K.goal_yrange[1]);
    period = rng_range(g, K.goal_time_period[0], K.goal_time_period[1]);
    mass[0] = rng_range(g, K.obj_mass_range[0], K.obj_mass_range[1]);
    mass[1] = rng_range(g, K.obj_mass_range[0], K.obj_mass_range[1]);
    if (K.beta_ball_mass_on)
      for (int k = 0; k < 2; ++k)
        mass[k] = rng_beta(g, K.beta_ball_mass[0], K.beta_ball_mass[1]) * (K.obj_mass_range[1] - K.obj_mass_range[0]) + K.obj_mass_range[0];
    for (int b = 0; b < 2; ++b)
      for (int k = 0; k < 3; ++k) {
        const double nominal = (double)M.geom_friction[3 * K.obj1_gid + k];
        fric[3 * b + k] = rng_range(g, nominal - K.obj_friction_change[k], nominal + K.obj_friction_change[k]);
      }
    size[0] = rng_range(g, K.obj_size_range[0], K.obj_size_range[1]);
    size[1] = rng_range(g, K.obj_size_range[0], K.obj_size_range[1]);
    if (K.beta_ball_size_on)
      for (int k = 0; k < 2; ++k)
        size[k] = rng_beta(g, K.beta_ball_size[0], K.beta_ball_size[1]) * (K.obj_size_range[1] - K.obj_size_range[0]) + K.obj_size_range[0];
    if (K.enable_rsi && philox_uniform(g) < K.rsi_probability) {
      do_rsi = 1;
      const double phase = rng_range(g, -MYO_PI, MYO_PI);
      start1 = 3.0 * MYO_PI / 4.0 + phase; start2 = -MYO_PI / 4.0 + phase;
    }
  }
  PHASE {
    if (lane == 0) {
      s.which_task = which; s.counter = 0; s.elapsed = 0; s.ep_ret = 0; s.ep_len = 0;
      s.start_angle[0] = start1; s.start_angle[1] = start2;
      s.x_radius = xr; s.y_radius = yr; s.time_period = period;
      s.ball_mass[0] = (T)mass[0]; s.ball_mass[1] = (T)mass[1];
      s.ball_size[0] = size[0]; s.ball_size[1] = size[1];
      for (int k = 0; k < 6; ++k) s.ball_fric[k] = (T)fric[k];
    }
  }
  SYNC();
  set_init_state(M, K, s, 0);
  if (do_rsi) {
    // self.step(np.zeros(39)); balls teleported onto the targets (xy), hand back to init pose
    task_step_core(M, K, s, (const float*)0);
    HP bx[4];
    const int nh = K.n_hand;
    bx[0] = s.target_w[0]; bx[1] = s.target_w[1]; bx[2] = s.target_w[3]; bx[3] = s.target_w[4];
    SYNC();
    set_init_state(M, K, s, 1);
    PHASE {
      if (lane == 0) {
        s.qpos[nh] = bx[0]; s.qpos[nh + 1] = bx[1]; s.qpos[nh + 7] = bx[2]; s.qpos[nh + 8] = bx[3];
      }
    }
    SYNC();
    if (K.kind == 2 && !K.balls_overlap) {
      const double a = rng_range(g, 0.0, 2.0 * MYO_PI);
      PHASE { if (lane == 0) { s.start_angle[0] = a; s.start_angle[1] = a - MYO_PI; } }
      SYNC();
    }
  }
  // noise (one scalar per joint group, exactly as the reference's broadcast assignments)
  if (K.kind == 1) {
    double nb[6] = {0, 0, 0, 0, 0, 0}, np_[3] = {0, 0, 0}, nf[3] = {0, 0, 0};
    if (K.noise_balls != 0) for (int k = 0; k < 6; ++k) nb[k] = rng_range(g, -K.noise_balls, K.noise_balls);
    if (K.noise_palm != 0) {
      np_[0] = rng_range(g, -MYO_PI / 2, -MYO_PI / 2 + MYO_PI / 18 * K.noise_palm);
      np_[1] = rng_range(g, -MYO_PI / 18 * K.noise_palm, MYO_PI / 18 * K.noise_palm);
      np_[2] = rng_range(g, -MYO_PI / 18 * K.noise_palm, MYO_PI / 18 * K.noise_palm);
    }
    if (K.noise_fingers != 0) {
      nf[0] = rng_range(g, -MYO_PI / 18 * K.noise_fingers, MYO_PI / 18 * K.noise_fingers);
      nf[1] = rng_range(g, 0.0, MYO_PI / 6 * K.noise_fingers);
      nf[2] = rng_range(g, -MYO_PI / 36 * K.noise_fingers, MYO_PI / 36 * K.noise_fingers);
    }
    PHASE {
      if (lane == 0) {
        const int nh = K.n_hand;
        if (K.noise_balls != 0) {
          const int idx[6] = {nh, nh + 1, nh + 2, nh + 7, nh + 8, nh + 9};
          for (int k = 0; k < 6; ++k) s.qpos[idx[k]] += nb[k];
        }
        if (K.noise_palm != 0) for (int k = 0; k < 3; ++k) s.qpos[k] = np_[k];
        if (K.noise_fingers != 0) {
          for (int k = 3; k < 7; ++k) s.qpos[k] = nf[0];
          for (int k = 7; k < nh; ++k) s.qpos[k] = ((k - 7) % 4 == 1) ? nf[2] : nf[1];
        }
      }
    }
    SYNC();
  } else if (K.noise_fingers != 0) {
    const double n0 = rng_range(g, -MYO_PI / 18 * K.noise_fingers, MYO_PI / 18 * K.noise_fingers);
    const double n1 = rng_range(g, 0.0, MYO_PI / 6 * K.noise_fingers);
    PHASE {
      if (lane == 0) {
        for (int k = 4; k < 7; ++k) s.qpos[k] = n0;
        for (int k = 7; k < K.n_hand; ++k) if ((k - 7) % 4 != 1) s.qpos[k] = n1;
      }
    }
    SYNC();
  }
  kinematics(M, s);
  baoding_obs_reward(M, K, s);
}

// ---- die reorient reset (reorient.py:124-181).  Draw order of the reference: goal position (3), the optional range choice per
// axis, the three Euler angles, the friction triple of every die geom, the size delta.  `enable_rsi` (:150-176) rewrites
// body_pos / body_quat of the Object body; that body carries the free joint, whose pose comes from qpos (robot.reset(init_qpos))
// and never from body_pos, so the reference's RSI branch leaves the state exactly as the plain reset does — and so does this.
template <typename T, int NC>
DEVFN void reorient_reset(const DevModel<T>& M_in, const TaskDev& K_in, Scratch<T, NC>& s_in, int env) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN
  Philox g;
  g.key0 = (unsigned int)K.seed; g.key1 = (unsigned int)(K.seed >> 32);
  g.c0 = (unsigned int)env; g.c1 = (unsigned int)s.episode; g.c2 = 0x44494531u; g.idx = 0;
  HP gp[3], e[3], q[4], lo[3], hi[3];
  for (int k = 0; k < 3; ++k) gp[k] = K.ro_goal_init_pos[k] + rng_range(g, K.ro_goal_pos[0], K.ro_goal_pos[1]);
  for (int ax = 0; ax < 3; ++ax) {
    lo[ax] = K.ro_goal_rot[0]; hi[ax] = K.ro_goal_rot[1];
    const int n = K.ro_n_rot_choice[ax];
    if (n > 0) {
      int c = (int)(philox_uniform(g) * (double)n);
      c = c >= n ? n - 1 : c;
      lo[ax] = K.ro_rot_choice[ax][c][0]; hi[ax] = K.ro_rot_choice[ax][c][1];
    }
  }
  for (int ax = 0; ax < 3; ++ax) e[ax] = rng_range(g, lo[ax], hi[ax]);
  ro_euler2quat(e, q);
  const int nf = 3 * (K.objg_gidn - K.objg_gid0);
  const unsigned int base = g.idx;
  g.idx = base + (unsigned int)nf;
  const HP del = rng_range(g, -K.ro_obj_size_change, K.ro_obj_size_change);
  PHASE {
    if (lane < nf) {    // one draw per (geom, coefficient): counter-based generator, lane j takes draw base + j
      Philox h = g;
      h.idx = base + (unsigned int)lane;
      const double nominal = (double)M.geom_friction[3 * K.objg_gid0 + lane], ch = K.obj_friction_change[lane % 3];
      const T f = (T)rng_range(h, nominal - ch, nominal + ch);
      constexpr int NF = Scratch<T, NC>::OBJG_NF;         // (the base scratch keeps the sliding coefficient only; the other two draws are consumed)
      if (lane % 3 < NF) S_OBJF(K, s)[(lane / 3) * NF + lane % 3] = f;
    }
    if (lane == 0) {
      s.which_task = 0; s.counter = 0; s.elapsed = 0; s.ep_ret = 0; s.ep_len = 0;
      for (int k = 0; k < 3; ++k) s.goal_pos[k] = gp[k];
      for (int k = 0; k < 4; ++k) s.goal_quat[k] = q[k];
      s.ball_size[0] = del;
    }
  }
  SYNC_G();      // (Scratch::SPILL: the friction triples went to the env's record in global memory)
  set_init_state(M, K, s, 0);
  kinematics(M, s);
  reorient_obs_reward(M, K, s);      // leaves pos_dist / rot_dist of the reset state (reorient.py:178-179)
}

template <typename T, int NC>
DEV void task_reset(const DevModel<T>& M, const TaskDev& K, Scratch<T, NC>& s, int env) {
  if (K.kind == MYO_TASK_REORIENT_K) reorient_reset(M, K, s, env); else baoding_reset(M, K, s, env);
}

// ---- HBM record <-> scratch
template <typename T, int NC>
DEV void load_env(const DevModel<T>& M_in, const TaskDev& K_in, const EnvRecordLayout L, double* rec, Scratch<T, NC>& s_in, int env, int pub = 0) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  if constexpr (sizeof(T) == sizeof(HP)) {      // fp64 stepper: warm start and controls stay in global memory (ScratchPoses<double>)
    // (the workspace of the hardware wave slot this workgroup runs in: myo_wave_slot, wave.h; MYO_WS_* in myo_physics.h)
    PHASE {
      if (lane == 0) {
        const size_t wsi = (size_t)myo_ws_acquire(K.slot_map, env, K.health);      // (given back by ws_release when the workgroup leaves the env)
        s.ws_idx = (int)wsi;
        s.warm_g = rec + L.off_warm; s.ctrl_g = K.ctrl_ws + wsi * MYO_ENVWS_N; s.tenj_g = s.ctrl_g + MYO_NU_MAX;
        if constexpr (Scratch<T, NC>::SPILL) s.big_g = K.big_ws + wsi * MYO_BIGWS_BYTES;      // (contact records, wrap results: CON / S_TWRES)
      }
    }
  }
  PHASE { if (lane == 0) s.pub = pub; }
  SYNC();
  PHASE {
    if (K.objg_gidn > 0 && !Scratch<T, NC>::SPILL) {            // (SPILL: read in place, S_OBJF)
      constexpr int NF = Scratch<T, NC>::OBJG_NF;
      for (int i = lane; i < NF * (K.objg_gidn - K.objg_gid0); i += 64) S_OBJF(K, s)[i] = (T)rec[L.off_objfric + 3 * (i / NF) + i % NF];
    }
    for (int i = lane; i < M.nq; i += 64) s.qpos[i] = rec[L.off_qpos + i];
    for (int i = lane; i < M.nv; i += 64) { s.qvel[i] = rec[L.off_qvel + i]; warm_set(s, i, (T)rec[L.off_warm + i]); }
    if constexpr (!Scratch<T, NC>::SPILL) { for (int i = lane; i < M.na; i += 64) s.act[i] = rec[L.off_act + i]; }      // (SPILL: read in place, S_ACT)
    for (int i = lane; i < M.nu; i += 64) ctrl_set(s, i, (T)0);
    if (lane < MYO_NV_MAX) s.hperm[lane] = (unsigned char)M.hperm[lane];
    if (lane == 0) {
      s.time = rec[L.off_time];
      const double* td = rec + L.off_taskd;
      for (int k = 0; k < MYO_TASKD_N; ++k) s.taskd[k] = td[k];
      const double* bd = rec + L.off_balld;
      if (K.objg_gidn <= 0) {       // (the ball slots share storage with the object group's friction)
        s.ball_mass[0] = (T)bd[0]; s.ball_mass[1] = (T)bd[1];
        for (int k = 0; k < 6; ++k) s.ball_fric[k] = (T)bd[2 + k];
      }
      s.ball_size[0] = bd[8]; s.ball_size[1] = bd[9];
      const double* mi = rec + L.off_misc;
      s.which_task = (int)mi[0]; s.counter = (int)mi[1]; s.elapsed = (int)mi[2]; s.episode = (int)mi[3];
      s.ep_ret = (T)mi[4]; s.ep_len = (int)mi[5];
      s.bad = (int)mi[6]; s.ncon = 0; s.nefc = 0; s.nl = 0; s.ntl = 0; s.solver_iter = 0;     // (mi[6] is non-zero only between the two parts of a split step)
    }
  }
  SYNC();
}

// the workspace block load_env took goes back (fp64 stepper; every path that called load_env ends here)
template <typename T, int NC>
DEV void ws_release(const TaskDev& K_in, Scratch<T, NC>& s_in) {
  MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  if constexpr (sizeof(T) == sizeof(HP)) {
    SYNC_G();      // (every lane's workspace stores are drained before lane 0 lets go)
    PHASE { if (lane == 0) myo_ws_release(K.slot_map, s.ws_idx); }
  }
}

template <typename T, int NC>
DEV void store_env(const DevModel<T>& M_in, const TaskDev& K_in, const EnvRecordLayout L, double* rec, const Scratch<T, NC>& s_in, int mid_step = 0) {
  MYO_BIND_M(T) MYO_BIND_K MYO_BIND_S(T)
  WAVE_FN_K
  // mid_step: 1 = the step goes on in another call (its state so far: `bad` is kept; the per-episode draws — ball / object data — have
  // not changed since load_env and are not stored again); | 2 = ... in another workgroup of this launch: write-through stores (st_pub,
  // wave.h); | 4 = this call did not start the step either: the task state (targets, counters) is as loaded
  const int wt = mid_step & 2, later = mid_step & 4;
  PHASE {
    if (K.kind == MYO_TASK_REORIENT_K && !mid_step && !Scratch<T, NC>::SPILL) {
      constexpr int NF = Scratch<T, NC>::OBJG_NF;
      for (int i = lane; i < NF * (K.objg_gidn - K.objg_gid0); i += 64) rec[L.off_objfric + 3 * (i / NF) + i % NF] = (double)S_OBJF(K, s)[i];
    }
    if constexpr (!Scratch<T, NC>::SPILL) {
      // qpos | qvel | act | warm start | time are ONE contiguous run of the record (myo_batch_create lays them out so): stored as one,
      // 64 consecutive doubles an instruction.  A part that hands its record on writes it through (sc1), and every store instruction
      // of a partial line is a memory-side request of its own: four arrays + five scalars were ~34 requests a hand-off, this is 19.
      const int n1 = M.nq, n2 = n1 + M.nv, n3 = n2 + M.na, n4 = n3 + M.nv;
      if constexpr (sizeof(T) == sizeof(HP)) {
        // fp64 stepper: its warm start is in GLOBAL memory (the wave slot's workspace), written by lane = dof at the end of the substep
        // just before: only THAT lane may read it back without a drain (wave.h, SYNC_G) — the run stops before it.  (Reading it from
        // the run's lanes made one 4000-step run in six end in another state: found by tools/dev/soak_repeat.sh.)
        for (int i = lane; i < n3; i += 64) {
          const double v = i < n1 ? (double)s.qpos[i] : (i < n2 ? (double)s.qvel[i - n1] : (double)s.act[i - n2]);
          st_pub(rec + L.off_qpos + i, v, wt);
        }
        for (int i = lane; i < M.nv; i += 64) st_pub(rec + L.off_warm + i, (double)warm_get(s, i), wt);
        if (lane == 0) st_pub(rec + L.off_time, (double)s.time, wt);
      } else {
        for (int i = lane; i <= n4; i += 64) {
          const double v = i < n1 ? (double)s.qpos[i] : (i < n2 ? (double)s.qvel[i - n1] : (i < n3 ? (double)s.act[i - n2] : (i < n4 ? (double)warm_get(s, i - n3) : (double)s.time)));
          st_pub(rec + L.off_qpos + i, v, wt);
        }
      }
    } else {
      for (int i = lane; i < M.nq; i += 64) st_pub(rec + L.off_qpos + i, (double)s.qpos[i], wt);
      for (int i = lane; i < M.nv; i += 64) { st_pub(rec + L.off_qvel + i, (double)s.qvel[i], wt); st_pub(rec + L.off_warm + i, (double)warm_get(s, i), wt); }
      if (lane == 0) st_pub(rec + L.off_time, (double)s.time, wt);
    }
    if (lane == 0) {
      if (!mid_step) {
        double* bd = rec + L.off_balld;
        if (K.objg_gidn <= 0) {
          bd[0] = (double)s.ball_mass[0]; bd[1] = (double)s.ball_mass[1];
          for (int k = 0; k < 6; ++k) bd[2 + k] = (double)s.ball_fric[k];
        }
        bd[8] = (double)s.ball_size[0]; bd[9] = (double)s.ball_size[1];
      }
    }
    if (lane < 7 && (lane == 6 || !later)) {      // the task counters, one lane each: one request instead of seven
      const double v = lane == 0 ? (double)s.which_task : (lane == 1 ? (double)s.counter : (lane == 2 ? (double)s.elapsed : (lane == 3 ? (double)s.episode :
                       (lane == 4 ? (double)s.ep_ret : (lane == 5 ? (double)s.ep_len : (mid_step ? (double)s.bad : 0.0))))));
      st_pub(rec + L.off_misc + lane, v, wt);
    }
    if (lane < MYO_TASKD_N && !later) st_pub(rec + L.off_taskd + lane, (double)s.taskd[lane], wt);
  }
  SYNC();
}

// ---- the three env-level entry points (one call = one env; the kernels are thin wrappers)
template <typename T, int RKM = -1, int NC>
DEV void env_step(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s,
                  int env, const float* act, float* obs, float* rew, unsigned char* done, unsigned char* trunc,
                  float* term_obs, float* comps, float* ep_info, unsigned char* bad_state, int k_lo = 0, int k_hi = -1, int pub = 0) {
  // pub: this call ends inside the step and another workgroup of the launch takes the record over (write-through stores, wave.h)
  WAVE_FN_K
  const int nobs = task_nobs(K, M.na);
  load_env(M, K, L, rec, s, env, pub);
  task_step_core<T, RKM>(M, K, s, act + (size_t)env * M.nu, k_lo, k_hi);
  if (k_hi >= 0 && k_hi < K.frame_skip) { store_env(M, K, L, rec, s, (s.pub ? 3 : 1) | (k_lo > 0 ? 4 : 0)); ws_release(K, s); return; }
  // A numerically blown-up env (mj_checkPos / mj_checkVel / mj_checkAcc: MuJoCo warns and resets the data) is not
  // an error of the batch: the env ends its episode with done = 1, reward 0, zero reward components except `done`,
  // is reset at once, and both the terminal and the returned observation are the (finite) reset observation, so that
  // no NaN reaches the normaliser statistics or the rollout buffer.  bad_state[env] = 1 tells the caller.
  const int bad = s.bad;
  const int fall = S_RWD(s)[6] != 0 || bad;
  int is_trunc = 0, is_done = fall;
  PHASE {
    if (lane == 0) { s.elapsed++; s.ep_len++; if (!bad) s.ep_ret += S_RWD(s)[7]; }
  }
  SYNC();
  if (s.elapsed >= K.max_episode_steps) { is_trunc = !fall; is_done = 1; }  // gym TimeLimit
  PHASE {
    if (lane == 0) {
      rew[env] = bad ? 0.0f : (float)S_RWD(s)[7];
      done[env] = (unsigned char)is_done;
      if (trunc) trunc[env] = (unsigned char)is_trunc;
      if (bad_state) bad_state[env] = (unsigned char)bad;
      if (ep_info) { ep_info[2 * env] = (float)s.ep_ret; ep_info[2 * env + 1] = (float)s.ep_len; }
    }
    if (comps && lane < 8) comps[(size_t)env * 8 + lane] = bad ? (lane == 6 ? 1.0f : 0.0f) : (float)S_RWD(s)[lane];
    if (term_obs && !bad) for (int i = lane; i < nobs; i += 64) term_obs[(size_t)env * nobs + i] = (float)S_OBS(s)[i];
  }
  SYNC();
  if (is_done) {
    PHASE { if (lane == 0) s.episode++; }
    SYNC();
    task_reset(M, K, s, env);
  }
  PHASE {
    for (int i = lane; i < nobs; i += 64) {
      obs[(size_t)env * nobs + i] = (float)S_OBS(s)[i];
      if (term_obs && bad) term_obs[(size_t)env * nobs + i] = (float)S_OBS(s)[i];
    }
  }
  SYNC();
  store_env(M, K, L, rec, s);
  ws_release(K, s);
}

// env.step(a) of the UNWRAPPED gym env for the envs selected by mask: no TimeLimit / Monitor accounting and
// no auto-reset — the steps MixtureModelBaodingEnv.reset takes with its base policy
// (/root/reference/src/envs/baoding.py:700-711).  done_out = the env's own `done` (ball dropped).
template <typename T, int RKM = -1, int NC>
DEV void env_step_inner(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s,
                        int env, const unsigned char* mask, const float* act, float* obs, unsigned char* done_out, int row = -1) {
  // row: the env's row in act / obs / done_out (compact form, myo_batch_step_inner_idx); -1: its own index
  WAVE_FN_K
  if (mask && !mask[env]) return;
  const int nobs = task_nobs(K, M.na);
  const int io = row < 0 ? env : row;
  load_env(M, K, L, rec, s, env);
  task_step_core<T, RKM>(M, K, s, act + (size_t)io * M.nu);
  const int bad = s.bad, fall = S_RWD(s)[6] != 0 || bad;
  if (bad) {                        // blown-up env: back to a finite reset state (see env_step)
    PHASE { if (lane == 0) s.episode++; }
    SYNC();
    task_reset(M, K, s, env);
  }
  PHASE {
    if (lane == 0 && done_out) done_out[io] = (unsigned char)fall;
    for (int i = lane; i < nobs; i += 64) obs[(size_t)io * nobs + i] = (float)S_OBS(s)[i];
  }
  SYNC();
  store_env(M, K, L, rec, s);
  ws_release(K, s);
}

template <typename T, int NC>
DEV void env_reset(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s,
                   int env, const unsigned char* mask, float* obs) {
  WAVE_FN_K
  if (mask && !mask[env]) return;
  const int nobs = task_nobs(K, M.na);
  load_env(M, K, L, rec, s, env);
  PHASE { if (lane == 0) s.episode++; }
  SYNC();
  task_reset(M, K, s, env);
  if (obs) { PHASE { for (int i = lane; i < nobs; i += 64) obs[(size_t)env * nobs + i] = (float)S_OBS(s)[i]; } SYNC(); }
  store_env(M, K, L, rec, s);
  ws_release(K, s);
}

template <typename T, int RKM = -1, int NC>
DEV void env_physics(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s,
                     int env, const double* ctrl, int nsub) {
  WAVE_FN_K
  load_env(M, K, L, rec, s, env);
  PHASE { for (int i = lane; i < M.nu; i += 64) ctrl_set(s, i, ctrl ? (T)ctrl[(size_t)env * M.nu + i] : (T)0); }
  SYNC();
  for (int k = 0; k < nsub; ++k) mj_step<T, RKM>(M, K, s);
  store_env(M, K, L, rec, s);
  ws_release(K, s);
}

// forward dynamics with intermediates exported (stage-wise parity tests)
// Census of the geom wraps (myo_batch_reset, myo_batch_tune_wrap_order): which sphere / cylinder wraps ENGAGE in the env's present state.
// cnt[k] counts position k of the wrap order (DevModel::gw_elem).  The wrap solver runs 64 wraps per pass in lockstep and its tangent solve
// is skipped by a pass none of whose wraps engages; k_wrap_reorder sorts the order by these counts so that the wraps that (almost) never
// engage — a fifth of the hand's 71 — share the last pass.  The env's record is not written.
template <typename T, int NC>
DEV void env_wrap_census(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s, int env, int* cnt) {
  WAVE_FN_K
  load_env(M, K, L, rec, s, env);
  kinematics(M, s);
  com_pos(M, K, s);
  tendon(M, K, s);
  for (int base = 0; base < M.ngw; base += 64) tendon_wrap_pass(M, K, s, base);
  auto wres = S_TWRES(s, M.nwrap);
  PHASE {
    for (int k = lane; k < M.ngw; k += 64)
      if (wres[7 * k] >= (HP)0) myo_count(cnt + k);
  }
  SYNC();
  ws_release(K, s);
}

struct DumpLayout {
  int ten_length, ten_J, M, qfrc_bias, qfrc_passive, qfrc_actuator, qacc_smooth, qacc, actuator_force, act_dot,
      counts, efc_aref, efc_D, site_xpos, subtree_com, xpos, total;
};
template <typename T, int NC>
DEV void env_forward_dump(const DevModel<T>& M, const TaskDev& K, const EnvRecordLayout& L, double* rec, Scratch<T, NC>& s,
                          int env, const double* ctrl, const DumpLayout& D, double* out_all) {
  WAVE_FN_K
  double* out = out_all + (size_t)env * D.total;
  load_env(M, K, L, rec, s, env);
  PHASE { for (int i = lane; i < M.nu; i += 64) ctrl_set(s, i, ctrl ? (T)ctrl[(size_t)env * M.nu + i] : (T)0); }
  SYNC();
  const int nv = M.nv;
  PHASE {
    for (int i = lane; i < D.total; i += 64) out[i] = 0;
  }
  SYNC();
  // same sequence as forward(); the force vectors share LDS with the system matrix, so they are
  // exported before the acceleration stage overwrites them
  kinematics(M, s);
  com_pos(M, K, s);
  tendon(M, K, s);
  for (int base = 0; base < M.ngw; base += 64) tendon_wrap_pass(M, K, s, base);
  for (int base = 0; base < M.nte; base += 64) tendon_element_pass(M, K, s, base);
  tendon_length_sums(M, s);
  crb(M, s);
  if (M.any_floss) friction_rows(M, K, s, 0);
  constraint_limits(M, K, s);
  if (M.any_floss) friction_rows(M, K, s, 1);
  if (M.any_gen) {
    for (int base = 0; base < M.npair_std; base += 64) collision_pass<true>(M, K, s, base);
    for (int base = M.npair_std; base < M.npair; base += 64) collision_pass_ext<true>(M, K, s, base - M.npair_std);
  } else {
    for (int base = 0; base < M.npair_std; base += 64) collision_pass<false>(M, K, s, base);
    for (int base = M.npair_std; base < M.npair; base += 64) collision_pass_ext<false>(M, K, s, base - M.npair_std);
  }
  contacts_clamp(K, s);
  // position-stage results first: the body poses (fp64 stepper) and the tendon lengths share LDS with vectors the later stages write
  PHASE {
    for (int t = lane; t < M.ntendon; t += 64) out[D.ten_length + t] = (double)S_TEN_LENGTH(s)[t];
    for (int sid = lane; sid < M.nsite; sid += 64) {
      HP p[3];
      body_point_hp(s, M.site_bodyid[sid], M.h_site_pos + 3 * sid, p);
      for (int k = 0; k < 3; ++k) out[D.site_xpos + 3 * sid + k] = p[k];
    }
    for (int b = lane; b < M.nbody; b += 64)
      for (int k = 0; k < 3; ++k) {
        out[D.subtree_com + 3 * b + k] = (double)S_COM(s)[3 * M.body_rootid[b] + k] + S_ORIGIN(s)[k];
        out[D.xpos + 3 * b + k] = (double)S_XPOS(s)[3 * b + k];
      }
  }
  SYNC();
  body_vectors(M, s, LOFF(s, S_QVELT(s)), LOFF(s, S_CVEL(s)));
  fwd_velocity(M, K, s);
  efc_reference(M, s);
  fwd_actuation(M, s);
  PHASE {
    for (int i = lane; i < nv; i += 64) {
      out[D.qfrc_bias + i] = (double)S_QFRC_BIAS(s)[i]; out[D.qfrc_passive + i] = (double)S_QFRC_PASSIVE(s)[i];
      out[D.qfrc_actuator + i] = (double)S_QFRC_ACTUATOR(s)[i];
    }
    if (lane == 0) {                          // (efc_jar holds aref until the solver starts; the padding rows of contact slots are skipped:
      int k = 0;                              //  the dump lists MuJoCo's rows)
      const int nlim_ = s.nl + s.ntl;
      for (int r = 0; r < s.nefc; ++r) {
        if (r >= nlim_ && con_pad(con_kind(CON(s, (r - nlim_) >> 2)), (r - nlim_) & 3)) continue;
        out[D.efc_aref + k++] = (double)S_AREF(s)[r];
      }
    }
    for (int i = lane; i < M.nu; i += 64) out[D.actuator_force + i] = (double)S_ACT_FORCE(s)[i];   // (lives in the solver's vectors)
  }
  SYNC();
  fwd_acceleration(M, s);
  PHASE {
    for (int t = lane; t < M.ntendon; t += 64) {
      unsigned long long m = M.tendon_dofmask[t];
      int slot = 0;
      while (m) { const int d = myo_ffsll(m); m &= m - 1; out[D.ten_J + t * nv + d] = (double)tenj_get(s, t, slot); slot++; }
    }
    for (int e = lane; e < M.nM; e += 64) {
      out[D.M + M.M_i[e] * nv + M.M_j[e]] = (double)s.qM[e];
      out[D.M + M.M_j[e] * nv + M.M_i[e]] = (double)s.qM[e];
    }
    for (int i = lane; i < nv; i += 64) {
      out[D.qacc_smooth + i] = (double)s.qacc_smooth[i];
      out[D.qacc + i] = (double)s.qacc[i];
    }
    for (int i = lane; i < M.na; i += 64) out[D.act_dot + i] = (double)S_ACT_DOT(s)[i];
    if (lane == 0) {
      // contacts and rows as MuJoCo counts them: a contact's first slot is kind 0 or 3, padding rows do not exist
      const int nlim_ = s.nl + s.ntl;
      int nc = 0, k = 0;
      for (int ci = 0; ci < s.ncon; ++ci) { const int kd = con_kind(CON(s, ci)); nc += (kd == 0 || kd == 3); }
      for (int r = 0; r < s.nefc; ++r) {
        if (r >= nlim_ && con_pad(con_kind(CON(s, (r - nlim_) >> 2)), (r - nlim_) & 3)) continue;
        out[D.efc_D + k++] = (double)row_D(s, r, nlim_);
      }
      out[D.counts] = nc; out[D.counts + 1] = k; out[D.counts + 2] = s.solver_iter; out[D.counts + 3] = s.nl;
    }
  }
  SYNC();
  ws_release(K, s);
}
