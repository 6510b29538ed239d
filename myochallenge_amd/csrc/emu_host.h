// emu_host.h — TEST TOOLING: the host side of the lane-serial emulation build (tests/emu/libmyobatch_emu.so = csrc/myobatch_emu.cpp).
// The kernel SOURCE (wave.h / myo_physics.h / myo_task.h, compiled with -DMYO_EMU: a phase's lanes run one after the other) is stepped
// env by env on the CPU, through the same records, layouts and task blocks as the product (csrc/myo_host.h), so that the wave-parallel
// algorithm can be checked against the oracle without a GPU.  It implements the env-path entry points of include/myobatch.h only;
// the PPO-side kernels have no CPU twin.  Nothing here is compiled into libmyobatch.so.
#pragma once
// ------------------------------------------------------------------------------------------ backend
typedef void* be_stream;
static int be_malloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : -1; }
static void be_free(void* p) { free(p); }
static int be_h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); return 0; }
static int be_set_device(int) { return 0; }
static const char* be_errstr(int) { return "emu"; }

static int be_batch_workspaces(myo_batch* b, int n_envs, int device) {
  (void)device;
  int rc = 0;
  void* w = nullptr;             // (emulation: one workspace per env)
  rc |= be_malloc(&w, sizeof(double) * (size_t)n_envs * MYO_ENVWS_N);
  b->K.ctrl_ws = (double*)w;
  if (w) b->allocs.push_back(w);
  if (b->ncap > MYO_NCON_MAX) {
    void* g = nullptr;
    rc |= be_malloc(&g, (size_t)n_envs * MYO_BIGWS_BYTES);
    b->K.big_ws = (char*)g;
    if (g) b->allocs.push_back(g);
  }
  return rc;
}
static int be_batch_launch_state(myo_batch* b, const myo_model* m, int n_envs, int rc) { (void)b; (void)m; (void)n_envs; return rc; }
static void be_batch_release(myo_batch* b, int device, int destroying) { (void)b; (void)device; (void)destroying; }

static void xfer(myo_batch* b, int off, int cnt, double* ext, int to_ext, be_stream st) {
  if (!ext) return;
  (void)st;
  for (int e = 0; e < b->n; ++e)
    for (int k = 0; k < cnt; ++k) {
      double* r = b->rec + (size_t)e * b->L.stride + off + k;
      if (to_ext) ext[(size_t)e * cnt + k] = *r; else *r = ext[(size_t)e * cnt + k];
    }
}
static void xfer_i(myo_batch* b, int off, int cnt, int* ext, int to_ext, be_stream st) {
  if (!ext) return;
  (void)st;
  for (int e = 0; e < b->n; ++e)
    for (int k = 0; k < cnt; ++k) {
      double* r = b->rec + (size_t)e * b->L.stride + off + k;
      if (to_ext) ext[(size_t)e * cnt + k] = (int)*r; else *r = (double)ext[(size_t)e * cnt + k];
    }
}

extern "C" int myo_batch_get_state(myo_batch* b, double* qpos, double* qvel, double* act, double* time, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer(b, b->L.off_qpos, b->nq, qpos, 1, st); xfer(b, b->L.off_qvel, b->nv, qvel, 1, st);
  xfer(b, b->L.off_act, b->na, act, 1, st); xfer(b, b->L.off_time, 1, time, 1, st);
  return MYO_OK;
}
extern "C" int myo_batch_set_state(myo_batch* b, const double* qpos, const double* qvel, const double* act,
                                   const double* time, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer(b, b->L.off_qpos, b->nq, (double*)qpos, 0, st); xfer(b, b->L.off_qvel, b->nv, (double*)qvel, 0, st);
  xfer(b, b->L.off_act, b->na, (double*)act, 0, st); xfer(b, b->L.off_time, 1, (double*)time, 0, st);
  return MYO_OK;
}
extern "C" int myo_batch_set_bad_state_buffer(myo_batch* b, uint8_t* bad_state) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  b->bad_state = bad_state;
  return MYO_OK;
}
extern "C" int myo_batch_warmstart(myo_batch* b, double* get_w, const double* set_w, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer(b, b->L.off_warm, b->nv, get_w, 1, st); xfer(b, b->L.off_warm, b->nv, (double*)set_w, 0, st);
  return MYO_OK;
}
extern "C" int myo_batch_set_task(myo_batch* b, const int32_t* task_i, const double* task_d, const double* ball_d, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer_i(b, b->L.off_misc, 2, (int*)task_i, 0, st);
  xfer(b, b->L.off_taskd, MYO_TASKD_N, (double*)task_d, 0, st); xfer(b, b->L.off_balld, MYO_BALLD_N, (double*)ball_d, 0, st);
  return MYO_OK;
}
extern "C" int myo_batch_set_object_group(myo_batch* b, int gid0, int gidn) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (!(gid0 == -1 && gidn == -1) && (gid0 < 0 || gidn <= gid0 || gidn > b->ngeom)) return fail(MYO_E_ARG, "bad geom range");
  if (b->K.kind != MYO_TASK_NONE) return fail(MYO_E_STATE, "object groups are for physics-only batches (the reorient task owns its own; the Baoding tasks have the two balls)");
  if (gidn - gid0 > MYO_OBJG_MAX) return fail(MYO_E_UNSUPPORTED, "an object group holds at most %d geoms", MYO_OBJG_MAX);
  b->K.objg_gid0 = gid0; b->K.objg_gidn = gidn;
  if (gidn > 0) {   // every env starts from the model's friction of the group's geoms
    const int cnt = 3 * (gidn - gid0);
    std::vector<double> host((size_t)b->n * cnt);
    for (int e = 0; e < b->n; ++e) for (int j = 0; j < cnt; ++j) host[(size_t)e * cnt + j] = b->geom_friction[3 * gid0 + j];
    void* tmp = nullptr;
    int rc = be_malloc(&tmp, host.size() * sizeof(double));
    if (!rc) rc = be_h2d(tmp, host.data(), host.size() * sizeof(double));
    if (rc) { if (tmp) be_free(tmp); return fail(MYO_E_DEVICE, "object group upload failed: %s", be_errstr(rc)); }
    xfer(b, b->L.off_objfric, cnt, (double*)tmp, 0, (be_stream)0);
    be_free(tmp);
  }
  return MYO_OK;
}
extern "C" int myo_batch_object_friction(myo_batch* b, const double* set_fric, double* get_fric, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (b->K.objg_gidn <= 0) return fail(MYO_E_STATE, "batch has no object group");
  be_stream st = (be_stream)stream;
  const int cnt = 3 * (b->K.objg_gidn - b->K.objg_gid0);
  xfer(b, b->L.off_objfric, cnt, (double*)set_fric, 0, st); xfer(b, b->L.off_objfric, cnt, get_fric, 1, st);
  return MYO_OK;
}
extern "C" int myo_batch_bind_constants(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  (void)stream;
  return MYO_OK;
}
extern "C" int myo_batch_get_task(myo_batch* b, int32_t* task_i, double* task_d, double* ball_d, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer_i(b, b->L.off_misc, 2, (int*)task_i, 1, st);
  xfer(b, b->L.off_taskd, MYO_TASKD_N, task_d, 1, st); xfer(b, b->L.off_balld, MYO_BALLD_N, ball_d, 1, st);
  return MYO_OK;
}

#define FOR_ENVS_T(TT, NCV, call) { Scratch<TT, NCV>* s = new Scratch<TT, NCV>(); memset(s, 0, sizeof *s); RkScratch<TT>* rk = new RkScratch<TT>(); s->rk = rk; for (int env = 0; env < b->n; ++env) { double* rec = b->rec + (size_t)env * b->L.stride; call; } delete s; delete rk; }
#define FOR_ENVS_F64(call) { if (b->ncap > MYO_NCON_MAX) FOR_ENVS_T(double, MYO_NCON_BIG, call) else FOR_ENVS_T(double, MYO_NCON_F64, call) }
#define FOR_ENVS_F32(call) { if (b->ncap > MYO_NCON_MAX) FOR_ENVS_T(float, MYO_NCON_BIG, call) else FOR_ENVS_T(float, MYO_NCON_MAX, call) }

extern "C" int myo_batch_tune_wrap_order(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  (void)stream;
  return MYO_OK;
}

extern "C" int myo_batch_reset(myo_batch* b, const uint8_t* mask, float* obs, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  (void)stream;
  if (b->dtype == MYO_F64) FOR_ENVS_F64(env_reset<double>(b->Md, b->K, b->L, rec, *s, env, mask, obs))
  else FOR_ENVS_F32(env_reset<float>(b->Mf, b->K, b->L, rec, *s, env, mask, obs))
  return MYO_OK;
}

extern "C" int myo_batch_step(myo_batch* b, const float* act, float* obs, float* rew, uint8_t* done, uint8_t* trunc,
                              float* term_obs, float* comps, float* ep_info, void* stream) {
  if (!b || !act || !obs || !rew || !done) return fail(MYO_E_ARG, "myo_batch_step: act/obs/rew/done are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  (void)stream;
  // (the parts of the step plan one after the other, each through the env record like the workgroups of k_step)
  for (int p = 0; p < b->plan.nparts; ++p) {
    const int k_lo = b->plan.k[p], k_hi = p == b->plan.nparts - 1 ? -1 : b->plan.k[p + 1];
    if (b->dtype == MYO_F64) FOR_ENVS_F64(env_step<double>(b->Md, b->K, b->L, rec, *s, env, act, obs, rew, done, trunc, term_obs, comps, ep_info, b->bad_state, k_lo, k_hi))
    else FOR_ENVS_F32(env_step<float>(b->Mf, b->K, b->L, rec, *s, env, act, obs, rew, done, trunc, term_obs, comps, ep_info, b->bad_state, k_lo, k_hi))
  }
  return MYO_OK;
}

// test hook: put the step plan's generation counter (and every env's state, consistently) at `gen` — the counter wraps after
// 2^28 steps and the protocol's arithmetic is modulo 2^32 (tests/test_step_parts.py steps across the wrap)
extern "C" int myo_batch_set_step_generation(myo_batch* b, unsigned int gen) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  (void)gen;
  return MYO_OK;
}

// Health counters of a batch (synchronises the device).  out[0]: k_step workgroups that found their env's hand-off state in another
// generation than the launch's (see the protocol comment at k_step); out[1]: substeps in which an env had more contacts than its
// scratch holds (the surplus was dropped); out[2]: the same for limit rows; out[3]: the most contact slots such a substep asked for.  All 0 in a healthy batch.
extern "C" int myo_batch_health(myo_batch* b, int out[4]) {
  if (!b || !out) return fail(MYO_E_ARG, "null argument");
  for (int k = 0; k < 4; ++k) out[k] = 0;
  if (!b->K.health) return MYO_OK;
  memcpy(out, b->K.health, 4 * sizeof(int));
  return MYO_OK;
}

// Device check of what the per-slot workspace rests on (myo_wave_slot, wave.h): `n_workgroups` one-wave workgroups with `lds_bytes` of
// dynamic LDS (k_step's footprint: the same residency) each take their slot's occupancy counter, stay for ~20 us, and leave.
// out[0]: workgroups that found their slot occupied (must be 0); out[1]: distinct slots seen; out[2]: largest slot index; out[3]: bit mask of XCC ids.
extern "C" int myo_debug_wave_slots(int device, int n_workgroups, int lds_bytes, int32_t out[4]) {
  if (!out || n_workgroups <= 0 || lds_bytes < 0 || lds_bytes > 65536) return fail(MYO_E_ARG, "bad argument");
  (void)device;
  return fail(MYO_E_UNSUPPORTED, "myo_debug_wave_slots: no wave slots in the emulation build");
}

extern "C" int myo_batch_step_inner(myo_batch* b, const uint8_t* mask, const float* act, float* obs, uint8_t* done, void* stream) {
  if (!b || !act || !obs) return fail(MYO_E_ARG, "myo_batch_step_inner: act/obs are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  (void)stream;
  if (b->dtype == MYO_F64) FOR_ENVS_F64(env_step_inner<double>(b->Md, b->K, b->L, rec, *s, env, mask, act, obs, done))
  else FOR_ENVS_F32(env_step_inner<float>(b->Mf, b->K, b->L, rec, *s, env, mask, act, obs, done))
  return MYO_OK;
}

// Compact form of myo_batch_step_inner: the envs idx[0 .. n_idx) (dev int32; -1 = empty slot) take one unwrapped env step with
// row r of act [n_idx, nu]; row r of obs [n_idx, obs_dim] and done [n_idx] (may be NULL) receive env idx[r]'s results.  An env
// must not be listed twice.  MixtureModelBaodingEnv's base phase runs the few envs that were just reset this way.
extern "C" int myo_batch_step_inner_idx(myo_batch* b, const int* idx, int n_idx, const float* act, float* obs, uint8_t* done, void* stream) {
  if (!b || !idx || !act || !obs || n_idx <= 0) return fail(MYO_E_ARG, "myo_batch_step_inner_idx: idx/act/obs are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  (void)stream;
  for (int r = 0; r < n_idx; ++r) {
    const int e = idx[r];
    if (e < 0 || e >= b->n) continue;
#define ONE_ENV(TT, NCV, MD) { Scratch<TT, NCV>* s = new Scratch<TT, NCV>(); memset(s, 0, sizeof *s); RkScratch<TT>* rk = new RkScratch<TT>(); s->rk = rk; \
      env_step_inner<TT>(MD, b->K, b->L, b->rec + (size_t)e * b->L.stride, *s, e, (const unsigned char*)nullptr, act, obs, done, r); delete s; delete rk; }
    if (b->dtype == MYO_F64) { if (b->ncap > MYO_NCON_MAX) ONE_ENV(double, MYO_NCON_BIG, b->Md) else ONE_ENV(double, MYO_NCON_F64, b->Md) }
    else { if (b->ncap > MYO_NCON_MAX) ONE_ENV(float, MYO_NCON_BIG, b->Mf) else ONE_ENV(float, MYO_NCON_MAX, b->Mf) }
#undef ONE_ENV
  }
  return MYO_OK;
}

// Whole env records from one batch into another (same model, same task kind): dst env dst_idx[r] <- src env src_idx[r], r < k.
// The record is everything an env is between two steps (state, warm start, task scalars, per-episode draws, counters), so the
// destination env continues exactly where the source env stood.  MixtureModelBaodingEnv hands pre-played episodes (reset + base
// phase, done in bulk on a pool batch) to the envs that have just finished.
extern "C" int myo_batch_copy_envs(myo_batch* dst, const int* dst_idx, const myo_batch* src, const int* src_idx, int k, void* stream) {
  if (!dst || !src || !dst_idx || !src_idx || k < 0) return fail(MYO_E_ARG, "myo_batch_copy_envs: null argument");
  if (dst->L.stride != src->L.stride || dst->nq != src->nq || dst->nv != src->nv || dst->na != src->na || dst->K.kind != src->K.kind || dst->device != src->device)
    return fail(MYO_E_ARG, "myo_batch_copy_envs: the two batches differ in model, task kind or device");
  if (k == 0) return MYO_OK;
  (void)stream;
  for (int r = 0; r < k; ++r) {
    const int d = dst_idx[r], s = src_idx[r];
    if (d < 0 || d >= dst->n || s < 0 || s >= src->n) continue;
    memcpy(dst->rec + (size_t)d * dst->L.stride, src->rec + (size_t)s * src->L.stride, sizeof(double) * (size_t)dst->L.stride);
  }
  return MYO_OK;
}

extern "C" int myo_batch_physics_step(myo_batch* b, const double* ctrl, int nsub, void* stream) {
  if (!b || nsub < 0) return fail(MYO_E_ARG, "bad arguments");
  (void)stream;
  if (b->dtype == MYO_F64) FOR_ENVS_F64(env_physics<double>(b->Md, b->K, b->L, rec, *s, env, ctrl, nsub))
  else FOR_ENVS_F32(env_physics<float>(b->Mf, b->K, b->L, rec, *s, env, ctrl, nsub))
  return MYO_OK;
}

extern "C" int myo_batch_forward_dump(myo_batch* b, const double* ctrl, double* out, void* stream) {
  if (!b || !out) return fail(MYO_E_ARG, "bad arguments");
  (void)stream;
  if (b->dtype == MYO_F64) FOR_ENVS_F64(env_forward_dump<double>(b->Md, b->K, b->L, rec, *s, env, ctrl, b->D, out))
  else FOR_ENVS_F32(env_forward_dump<float>(b->Mf, b->K, b->L, rec, *s, env, ctrl, b->D, out))
  return MYO_OK;
}

extern "C" int myo_batch_enable_timing(myo_batch* b, int on) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  b->timing = on;
  return MYO_OK;
}
extern "C" double myo_batch_kernel_ms(myo_batch* b) {
  if (!b) return -1.0;
  return -1.0;
}


