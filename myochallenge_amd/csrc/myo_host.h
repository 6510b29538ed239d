// myo_host.h — the host side both builds of libmyobatch share: error reporting, the host model (blob -> tables), the batch
// (records, layouts, task block) and myo_batch_create / _destroy.  Included by csrc/myobatch.hip (the product: the HIP backend, the
// kernels and every entry point) and by csrc/myobatch_emu.cpp (test tooling: csrc/emu_host.h runs the same kernel SOURCE lane by lane
// on the CPU).  The including file defines, before the include:
//   MYO_BACKEND_NAME           "gfx950" | "MYO_EMU lane-serial test build"   (myo_version)
//   MYO_BACKEND_DENSE_NEWTON   0 | 1: factor the Newton system densely (no block-arrow tables)
//   MYO_BACKEND_BATCH_FIELDS   members the backend keeps in struct myo_batch
// and, after it, the backend functions declared below.
#pragma once
#include "../../include/myobatch.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/myo_model_blob.h"
#include "myo_mjb.h"
#include "myo_task.h"


// ------------------------------------------------------------------------------------------ backend (defined by the including file)
struct myo_batch;
struct myo_model;
static int be_malloc(void** p, size_t n);
static void be_free(void* p);
static int be_h2d(void* d, const void* h, size_t n);
static int be_set_device(int dev);
static const char* be_errstr(int e);
static int be_batch_workspaces(myo_batch* b, int n_envs, int device);                      // the fp64 stepper's workspaces (TaskDev::ctrl_ws / big_ws); returns error bits
static int be_batch_launch_state(myo_batch* b, const myo_model* m, int n_envs, int rc);    // what the launches need beside the records; returns rc | its own error bits
static void be_batch_release(myo_batch* b, int device, int destroying);                    // ... and their release (destroying = 0: a failed myo_batch_create)

static thread_local char g_err[4096] = "";      // (room for a loader report that lists every unsupported feature of a model)
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* myo_last_error(void) { return g_err; }
#ifndef MYO_BUILD_ID
#define MYO_BUILD_ID "unknown"
#endif
extern "C" const char* myo_version(void) {      // "... build <hash of the native sources>" (myochallenge_amd/build.py:source_id)
  return "myobatch 0.1 (" MYO_BACKEND_NAME ") build " MYO_BUILD_ID;
}

// ------------------------------------------------------------------------------------------ host model
struct myo_model {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, npair, nM, maxdepth;
  int integrator, iterations, disableflags, any_damping, any_tendon_passive, nlead, ngw, nte, npair_std, ld_nfq, ld_nsq, arrow_nf, any_rot, any_gen, any_floss;
  unsigned long long arrow_pad;
  double timestep, tolerance, impratio, gravity[3], meaninertia;
#define X(n) std::vector<int> n;
  MYO_MODEL_INT_ARRAYS(X)
#undef X
#define X(n) std::vector<unsigned long long> n;
  MYO_MODEL_U64_ARRAYS(X)
#undef X
#define X(n) std::vector<double> n;
  MYO_MODEL_REAL_ARRAYS(X)
#undef X
};

static const myo_blob_field* blob_find(const void* blob, const char* name) {
  const myo_blob_header* h = (const myo_blob_header*)blob;
  const myo_blob_field* f = (const myo_blob_field*)((const char*)blob + sizeof(myo_blob_header));
  for (uint32_t i = 0; i < h->n_fields; ++i)
    if (strncmp(f[i].name, name, MYO_BLOB_NAME_LEN) == 0) return &f[i];
  return nullptr;
}
static bool get_i(const void* blob, size_t nbytes, const char* name, std::vector<int>& out) {
  const myo_blob_field* f = blob_find(blob, name);
  if (!f || f->dtype != MYO_BLOB_I32 || f->offset + 4ull * f->count > nbytes) return false;
  const int* p = (const int*)((const char*)blob + f->offset);
  out.assign(p, p + f->count);
  return true;
}
static bool get_d(const void* blob, size_t nbytes, const char* name, std::vector<double>& out) {
  const myo_blob_field* f = blob_find(blob, name);
  if (!f || f->dtype != MYO_BLOB_F64 || f->offset + 8ull * f->count > nbytes) return false;
  const double* p = (const double*)((const char*)blob + f->offset);
  out.assign(p, p + f->count);
  return true;
}
static void quat2mat_h(const double* q, double* R) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}

// Item tables of the level-parallel tree-sparse L'DL (csrc/myo_sparse_ldl.h; mj_factorM / mj_solveM order of operations per
// entry, scheduled by depth in the dof tree).  ld_fac word: e_ij | e_ki << 8 | e_kj << 16 | e_kk << 24 (indices into qM's
// dof_Madr layout) for  M[i,j] -= M[k,i] M[k,j] / M[k,k];  ld_sol word: e_ij | i << 8 | j << 16, j a strict ancestor of i.
// Levels deepest first, each padded to whole 64-item chunks with no-ops (e_ij = 255).  Rows >= nlead are diagonal by
// construction and get no items.  A tree that needs more chunks than the kernel preloads keeps the dense solves.
static void build_ldl_tables(myo_model* m) {
  const int nv = m->nv;
  std::vector<int> depth(nv, 0);
  int maxd = 0;
  for (int i = 0; i < nv; ++i) { depth[i] = m->dof_parentid[i] < 0 ? 0 : depth[m->dof_parentid[i]] + 1; if (depth[i] > maxd) maxd = depth[i]; }
  std::vector<int> fac, sol;
  const int nop = 255;
  for (int L = maxd; L >= 1; --L) {
    size_t f0 = fac.size(), s0 = sol.size();
    for (int k = 0; k < nv && k < m->nlead; ++k) {
      if (depth[k] != L) continue;
      std::vector<int> anc;                               // strict ancestors of k, nearest first
      for (int a = m->dof_parentid[k]; a >= 0; a = m->dof_parentid[a]) anc.push_back(a);
      const int ek = m->dof_Madr[k];
      for (size_t p = 0; p < anc.size(); ++p) {
        sol.push_back((ek + 1 + (int)p) | (k << 8) | (anc[p] << 16));
        for (size_t q = p; q < anc.size(); ++q)           // i = anc[p], j = anc[q] (j = i or an ancestor of i)
          fac.push_back((m->dof_Madr[anc[p]] + (int)(q - p)) | ((ek + 1 + (int)p) << 8) | ((ek + 1 + (int)q) << 16) | (ek << 24));
      }
    }
    if (fac.size() > f0) while (fac.size() % 64) fac.push_back(nop);
    if (sol.size() > s0) while (sol.size() % 64) sol.push_back(nop);
  }
  m->ld_nfq = (int)fac.size() / 64; m->ld_nsq = (int)sol.size() / 64;
  if (m->ld_nfq > MYO_LD_FQ || m->ld_nsq > MYO_LD_SQ || m->nM >= nop || getenv("MYO_DENSE_MSOLVE")) { m->ld_nfq = 0; m->ld_nsq = -1; fac.clear(); sol.clear(); }   // ld_nsq < 0: dense
  fac.resize((size_t)MYO_LD_FQ * 64, nop); sol.resize((size_t)MYO_LD_SQ * 64, nop);
  m->ld_fac = fac; m->ld_sol = sol;
}

// Block-arrow structure of the Newton system H = M + J'DJ (csrc/myo_arrow_chol.h).  Leaf blocks: subtrees of the dof tree with at
// most MYO_ARROW_B dofs that no constraint couples to another block — M couples a dof with its ancestors and descendants only, a
// tendon-limit row couples the dofs its tendon moves, a contact the dofs that move either body (the hand: five 4-dof fingers; the
// wrist and the free balls / die couple everything and form the separator).  hperm[dof] = row of the permuted system: separator
// rows 0..15, block f at 16 + 4 f; rows without a dof are identity (arrow_pad).  M_pkh[e] = packed-H offset of M's entry e in that
// order.  arrow_nf = 0 (structure absent, separator too large, MYO_DENSE_NEWTON set, emulation build): identity order, dense path.
static void build_arrow_tables(myo_model* m) {
  const int nv = m->nv;
  m->hperm.assign(MYO_NV_MAX, 0);
  for (int d = 0; d < MYO_NV_MAX; ++d) m->hperm[d] = d;
  m->arrow_nf = 0; m->arrow_pad = 0;
  auto finish = [&]() {
    m->M_pkh.assign(MYO_NM_MAX, 0);
    for (int e = 0; e < m->nM; ++e) {
      const int a = m->hperm[m->M_i[e]], b = m->hperm[m->M_j[e]], hi = a > b ? a : b, lo = a > b ? b : a;
      const int q = hi >> 2;
      m->M_pkh[e] = ((q * (q + 1)) << 3) + (((hi & 3) * (q + 1)) << 2) + lo;      // MYO_HIDX(hi, lo)
    }
    if (!m->arrow_nf) for (int d = nv; d < MYO_NV_MAX; ++d) m->arrow_pad |= 1ull << d;
  };
  if (MYO_BACKEND_DENSE_NEWTON) { finish(); return; }      // (the lane-serial emulation factors H densely: no MFMA there)
  if (getenv("MYO_DENSE_NEWTON") || nv > MYO_NV_MAX) { finish(); return; }
  // subtree sizes; candidate blocks = maximal subtrees of <= MYO_ARROW_B dofs
  std::vector<int> size(nv, 1), block(nv, -1);
  for (int d = nv - 1; d >= 0; --d) if (m->dof_parentid[d] >= 0) size[m->dof_parentid[d]] += size[d];
  std::vector<std::vector<int>> blocks;
  for (int d = 0; d < nv; ++d) {
    const int par = m->dof_parentid[d];
    if (size[d] <= MYO_ARROW_B && (par < 0 || size[par] > MYO_ARROW_B)) {
      std::vector<int> mem;
      for (int e = d; e < nv; ++e) { int a = e; while (a >= 0 && a != d) a = m->dof_parentid[a]; if (a == d) { mem.push_back(e); block[e] = (int)blocks.size(); } }
      blocks.push_back(mem);
    }
  }
  // constraint couplings between blocks; the most-coupled block goes to the separator until none is left (a free ball's
  // trailing dofs form a candidate block that every finger touches: it goes, the fingers stay)
  std::vector<unsigned long long> sets;
  for (int t = 0; t < m->ntendon; ++t) sets.push_back(m->tendon_dofmask[t]);
  for (int p = 0; p < m->npair; ++p) sets.push_back(m->pc_mask[2 * (size_t)p] | m->pc_mask[2 * (size_t)p + 1]);
  const size_t nblk = blocks.size();
  std::vector<std::vector<char>> adj(nblk, std::vector<char>(nblk, 0));
  for (unsigned long long mk : sets) {
    std::vector<int> bs;
    for (int d = 0; d < nv; ++d) if (((mk >> d) & 1ull) && block[d] >= 0 && std::find(bs.begin(), bs.end(), block[d]) == bs.end()) bs.push_back(block[d]);
    for (int a : bs) for (int b : bs) if (a != b) adj[a][b] = 1;
  }
  std::vector<char> demote(nblk, 0);
  for (;;) {
    int worst = -1, wdeg = 0;
    for (size_t a = 0; a < nblk; ++a) {
      if (demote[a]) continue;
      int deg = 0;
      for (size_t b = 0; b < nblk; ++b) if (!demote[b] && adj[a][b]) deg++;
      if (deg > wdeg) { wdeg = deg; worst = (int)a; }
    }
    if (worst < 0) break;
    demote[worst] = 1;
  }
  // keep the largest blocks (at most MYO_ARROW_NF); everything else is separator
  std::vector<int> keep;
  for (size_t b = 0; b < blocks.size(); ++b) if (!demote[b]) keep.push_back((int)b);
  std::stable_sort(keep.begin(), keep.end(), [&](int a, int b) { return blocks[a].size() > blocks[b].size(); });
  if ((int)keep.size() > MYO_ARROW_NF) keep.resize(MYO_ARROW_NF);
  std::vector<char> in_keep(blocks.size(), 0);
  for (int b : keep) in_keep[b] = 1;
  int nsep = 0;
  for (int d = 0; d < nv; ++d) if (block[d] < 0 || !in_keep[block[d]]) nsep++;
  if (keep.size() < 2 || nsep > MYO_ARROW_S) { finish(); return; }
  std::sort(keep.begin(), keep.end());
  unsigned long long used = 0;
  int ns = 0;
  for (int d = 0; d < nv; ++d) if (block[d] < 0 || !in_keep[block[d]]) { m->hperm[d] = ns; used |= 1ull << ns; ns++; }
  for (size_t f = 0; f < keep.size(); ++f)
    for (size_t t = 0; t < blocks[keep[f]].size(); ++t) { const int r = MYO_ARROW_S + MYO_ARROW_B * (int)f + (int)t; m->hperm[blocks[keep[f]][t]] = r; used |= 1ull << r; }
  m->arrow_nf = (int)keep.size();
  m->arrow_pad = ~used & ((1ull << MYO_NV_MAX) - 1ull);
  finish();
}

// what sol_param (myo_physics.h) needs of a (solref[2], solimp[5]) pair, resolved once: ref2 <- (K, B) with refsafe applied; imp5 <- (d0, d1,
// 1 / width — 0 when the impedance does not depend on the position —, midpoint, power), clamped as MuJoCo's getsolparam / getimpedance clamp them
static void sol_precompute(double* ref2, double* imp5, double timestep, int disableflags) {
  auto clamp = [](double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); };
  const double d0 = clamp(imp5[0], 0.0001, 0.9999), d1 = clamp(imp5[1], 0.0001, 0.9999);
  const double width = imp5[2] < 0 ? 0.0 : imp5[2];
  const double mid = clamp(imp5[3], 0.0001, 0.9999), power = imp5[4] < 1 ? 1.0 : imp5[4];
  double tc = ref2[0];
  const double dr = ref2[1];
  double K, B;
  if (tc > 0) {
    if (!(disableflags & (1 << 11)) && tc < 2 * timestep) tc = 2 * timestep;
    K = 1.0 / std::max(1e-15, d1 * d1 * tc * tc * dr * dr);
    B = 2.0 / std::max(1e-15, d1 * tc);
  } else { K = -tc / (d1 * d1); B = -dr / d1; }
  ref2[0] = K; ref2[1] = B;
  imp5[0] = d0; imp5[1] = d1; imp5[2] = (d0 == d1 || width <= 1e-15) ? 0.0 : 1.0 / width; imp5[3] = mid; imp5[4] = power;
}

static int model_from_blob_impl(const void* blob, size_t nbytes, myo_model** out);
extern "C" int myo_model_from_blob(const void* blob, size_t nbytes, myo_model** out) {
  try { return model_from_blob_impl(blob, nbytes, out); }          // no C++ exception crosses the C ABI
  catch (const std::exception& e) { return fail(MYO_E_ARG, "model blob rejected: %s", e.what()); }
}
static int model_from_blob_impl(const void* blob, size_t nbytes, myo_model** out) {
  if (!blob || !out || nbytes < sizeof(myo_blob_header)) return fail(MYO_E_ARG, "null or short blob");
  const myo_blob_header* h = (const myo_blob_header*)blob;
  if (h->magic != MYO_BLOB_MAGIC || h->version != MYO_BLOB_VERSION || h->total_bytes != nbytes ||
      sizeof(myo_blob_header) + (size_t)h->n_fields * sizeof(myo_blob_field) > nbytes)
    return fail(MYO_E_ARG, "bad model blob header");
  myo_model* m = new myo_model();
  std::vector<int> sizes, opt_i, trntype, trnid, body_weldid, geom_condim, pair_sub, pair_xp, xp_dim;
  std::vector<double> opt_d, body_iquat, geom_quat, xp_margin, xp_gap, xp_solref, xp_solimp, xp_friction;
  bool ok = get_i(blob, nbytes, "sizes", sizes) && sizes.size() >= 10 && get_i(blob, nbytes, "opt_int", opt_i) &&
            opt_i.size() >= 4 && get_d(blob, nbytes, "opt_f64", opt_d) && opt_d.size() >= 8;
  const char* missing = nullptr;
#define GI(n) if (ok && !get_i(blob, nbytes, #n, m->n)) { ok = false; missing = #n; }
#define GD(n) if (ok && !get_d(blob, nbytes, #n, m->n)) { ok = false; missing = #n; }
  GI(body_parentid) GI(body_rootid) GI(body_jntnum) GI(body_jntadr) GI(body_dofnum) GI(body_dofadr)
  GI(jnt_type) GI(jnt_qposadr) GI(jnt_dofadr) GI(jnt_bodyid) GI(jnt_limited) GI(dof_bodyid) GI(dof_jntid)
  GI(dof_parentid) GI(geom_type) GI(geom_bodyid) GI(geom_priority) GI(site_bodyid) GI(tendon_adr)
  GI(tendon_num) GI(tendon_limited) GI(wrap_type) GI(wrap_objid) GI(actuator_dyntype) GI(actuator_gaintype)
  GI(actuator_biastype) GI(actuator_ctrllimited) GI(actuator_forcelimited)
  GD(qpos0) GD(qpos_spring) GD(body_pos) GD(body_quat) GD(body_ipos) GD(body_mass) GD(body_inertia)
  GD(body_invweight0) GD(jnt_solref) GD(jnt_solimp) GD(jnt_pos) GD(jnt_axis) GD(jnt_stiffness) GD(jnt_range)
  GD(jnt_margin) GD(dof_armature) GD(dof_damping) GD(dof_invweight0) GD(geom_solmix) GD(geom_solref)
  GD(geom_solimp) GD(geom_size) GD(geom_rbound) GD(geom_pos) GD(geom_friction) GD(geom_margin) GD(geom_gap)
  GD(site_pos) GD(tendon_solref_lim) GD(tendon_solimp_lim) GD(tendon_range) GD(tendon_margin)
  GD(tendon_stiffness) GD(tendon_damping) GD(tendon_lengthspring) GD(tendon_invweight0) GD(wrap_prm)
  GD(actuator_dynprm) GD(actuator_gainprm) GD(actuator_biasprm) GD(actuator_ctrlrange) GD(actuator_forcerange)
  GD(actuator_gear) GD(actuator_acc0) GD(actuator_lengthrange)
#undef GI
#undef GD
  if (ok) ok = get_i(blob, nbytes, "actuator_trntype", trntype) && get_i(blob, nbytes, "actuator_trnid", trnid) &&
               get_d(blob, nbytes, "body_iquat", body_iquat) && get_d(blob, nbytes, "geom_quat", geom_quat) &&
               get_i(blob, nbytes, "x_pair_geom1", m->pair_geom1) && get_i(blob, nbytes, "x_pair_geom2", m->pair_geom2);
  if (ok && !get_i(blob, nbytes, "x_pair_sub", pair_sub)) pair_sub.assign(m->pair_geom1.size(), 0);      // (older blobs: no box-box candidates)
  if (ok && !get_i(blob, nbytes, "geom_condim", geom_condim)) geom_condim.clear();                        // (older blobs: condim 3 everywhere)
  if (ok) {                                   // explicit <contact><pair> parameters (older blobs: none)
    if (!get_i(blob, nbytes, "x_pair_explicit", pair_xp) || pair_xp.size() != m->pair_geom1.size()) pair_xp.assign(m->pair_geom1.size(), -1);
    const bool have = get_i(blob, nbytes, "x_xp_dim", xp_dim) && get_d(blob, nbytes, "x_xp_margin", xp_margin) && get_d(blob, nbytes, "x_xp_gap", xp_gap) &&
                      get_d(blob, nbytes, "x_xp_solref", xp_solref) && get_d(blob, nbytes, "x_xp_solimp", xp_solimp) && get_d(blob, nbytes, "x_xp_friction", xp_friction);
    const size_t nx = have ? xp_dim.size() : 0;
    if (have && (xp_margin.size() != nx || xp_gap.size() != nx || xp_solref.size() != 2 * nx || xp_solimp.size() != 5 * nx || xp_friction.size() != 3 * nx)) {
      int rc = fail(MYO_E_ARG, "corrupt model: explicit contact pair arrays differ in length"); delete m; return rc;
    }
    for (int& v : pair_xp) if (v < -1 || v >= (int)nx) { int rc = fail(MYO_E_ARG, "corrupt model: explicit contact pair index %d", v); delete m; return rc; }
  }
  if (ok) {                                   // friction loss: optional fields (older blobs: none), MuJoCo's default solver parameters
    const size_t nv0 = sizes[1] > 0 && sizes[1] <= 4096 ? sizes[1] : 0, nt0 = sizes[8] > 0 && sizes[8] <= 4096 ? sizes[8] : 0;     // (untrusted sizes: capacity checks follow)
    auto opt = [&](const char* name, std::vector<double>& v, size_t cnt, std::initializer_list<double> def) {
      if (!get_d(blob, nbytes, name, v) || v.size() < cnt * def.size()) { v.clear(); for (size_t i = 0; i < cnt; ++i) v.insert(v.end(), def); }
    };
    opt("dof_frictionloss", m->dof_frictionloss, nv0, {0.0}); opt("dof_solref", m->dof_solref, nv0, {0.02, 1.0});
    opt("dof_solimp", m->dof_solimp, nv0, {0.9, 0.95, 0.001, 0.5, 2.0});
    opt("tendon_frictionloss", m->tendon_frictionloss, nt0, {0.0}); opt("tendon_solref_fri", m->tendon_solref_fri, nt0, {0.02, 1.0});
    opt("tendon_solimp_fri", m->tendon_solimp_fri, nt0, {0.9, 0.95, 0.001, 0.5, 2.0});
  }
  if (!ok) {
    int rc = fail(MYO_E_ARG, "model blob lacks field %s", missing ? missing : "(sizes/opt/derived)");
    delete m;
    return rc;
  }
  m->nq = sizes[0]; m->nv = sizes[1]; m->nu = sizes[2]; m->na = sizes[3]; m->nbody = sizes[4]; m->njnt = sizes[5];
  m->ngeom = sizes[6]; m->nsite = sizes[7]; m->ntendon = sizes[8]; m->nwrap = sizes[9];
  m->npair = (int)m->pair_geom1.size();
  m->npair_std = 0;
  m->integrator = opt_i[0]; m->iterations = opt_i[2]; m->disableflags = opt_i[3];
  m->timestep = opt_d[0]; m->tolerance = opt_d[1]; m->impratio = opt_d[2];
  m->gravity[0] = opt_d[3]; m->gravity[1] = opt_d[4]; m->gravity[2] = opt_d[5]; m->meaninertia = opt_d[7];
  // ---- the blob is untrusted input: sizes, array lengths and every id are checked before anything is indexed with them
  {
    char why[160] = "";
#define BAD(...) { snprintf(why, sizeof why, __VA_ARGS__); int rc = fail(MYO_E_ARG, "corrupt model: %s", why); delete m; return rc; }
    for (int k = 0; k < 10; ++k) if (sizes[k] < 0) BAD("negative size %d", k)
    if (m->nbody < 1 || m->na > m->nu) BAD("nbody < 1 or na > nu")
    if (m->pair_geom1.size() != m->pair_geom2.size() || pair_sub.size() != m->pair_geom1.size()) BAD("pair arrays differ in length")
    for (int v : pair_sub) if (v < 0 || v > 17) BAD("collision pair sub-index %d", v)
#define NEED(arr, cnt) if (m->arr.size() < (size_t)(cnt)) BAD("array %s has %zu entries, needs %zu", #arr, m->arr.size(), (size_t)(cnt))
    const size_t nb_ = m->nbody, nj_ = m->njnt, nv_ = m->nv, ng_ = m->ngeom, ns_ = m->nsite, nt_ = m->ntendon, nw_ = m->nwrap, nu_ = m->nu;
    NEED(body_parentid, nb_) NEED(body_rootid, nb_) NEED(body_jntnum, nb_) NEED(body_jntadr, nb_) NEED(body_dofnum, nb_) NEED(body_dofadr, nb_)
    NEED(jnt_type, nj_) NEED(jnt_qposadr, nj_) NEED(jnt_dofadr, nj_) NEED(jnt_bodyid, nj_) NEED(jnt_limited, nj_)
    NEED(dof_bodyid, nv_) NEED(dof_jntid, nv_) NEED(dof_parentid, nv_) NEED(geom_type, ng_) NEED(geom_bodyid, ng_) NEED(geom_priority, ng_)
    NEED(site_bodyid, ns_) NEED(tendon_adr, nt_) NEED(tendon_num, nt_) NEED(tendon_limited, nt_) NEED(wrap_type, nw_) NEED(wrap_objid, nw_)
    NEED(actuator_dyntype, nu_) NEED(actuator_gaintype, nu_) NEED(actuator_biastype, nu_) NEED(actuator_ctrllimited, nu_) NEED(actuator_forcelimited, nu_)
    NEED(qpos0, m->nq) NEED(qpos_spring, m->nq) NEED(body_pos, 3 * nb_) NEED(body_quat, 4 * nb_) NEED(body_ipos, 3 * nb_) NEED(body_mass, nb_)
    NEED(body_inertia, 3 * nb_) NEED(body_invweight0, 2 * nb_) NEED(jnt_solref, 2 * nj_) NEED(jnt_solimp, 5 * nj_) NEED(jnt_pos, 3 * nj_)
    NEED(jnt_axis, 3 * nj_) NEED(jnt_stiffness, nj_) NEED(jnt_range, 2 * nj_) NEED(jnt_margin, nj_) NEED(dof_armature, nv_) NEED(dof_damping, nv_)
    NEED(dof_invweight0, nv_) NEED(geom_solmix, ng_) NEED(geom_solref, 2 * ng_) NEED(geom_solimp, 5 * ng_) NEED(geom_size, 3 * ng_)
    NEED(geom_rbound, ng_) NEED(geom_pos, 3 * ng_) NEED(geom_friction, 3 * ng_) NEED(geom_margin, ng_) NEED(geom_gap, ng_) NEED(site_pos, 3 * ns_)
    NEED(tendon_solref_lim, 2 * nt_) NEED(tendon_solimp_lim, 5 * nt_) NEED(tendon_range, 2 * nt_) NEED(tendon_margin, nt_) NEED(tendon_stiffness, nt_)
    NEED(tendon_damping, nt_) NEED(tendon_lengthspring, nt_) NEED(tendon_invweight0, nt_) NEED(wrap_prm, nw_) NEED(actuator_dynprm, 10 * nu_)
    NEED(actuator_gainprm, 10 * nu_) NEED(actuator_biasprm, 10 * nu_) NEED(actuator_ctrlrange, 2 * nu_) NEED(actuator_forcerange, 2 * nu_)
    NEED(actuator_gear, 6 * nu_) NEED(actuator_acc0, nu_) NEED(actuator_lengthrange, 2 * nu_)
#undef NEED
    if (geom_condim.empty()) geom_condim.assign(ng_, 3);
    if (geom_condim.size() < ng_) BAD("geom_condim too short")
    if (trntype.size() < nu_ || trnid.size() < 2 * nu_ || body_iquat.size() < 4 * nb_ || geom_quat.size() < 4 * ng_) BAD("actuator_trn* / body_iquat / geom_quat too short")
    if (m->body_parentid[0] != 0) BAD("body_parentid[0] != 0")
    for (int b = 0; b < m->nbody; ++b) {
      if (b > 0 && (m->body_parentid[b] < 0 || m->body_parentid[b] >= b)) BAD("body_parentid[%d] = %d", b, m->body_parentid[b])
      if (m->body_rootid[b] < 0 || m->body_rootid[b] >= m->nbody) BAD("body_rootid[%d] = %d", b, m->body_rootid[b])
      const int jn = m->body_jntnum[b], ja = m->body_jntadr[b], dn = m->body_dofnum[b], da = m->body_dofadr[b];
      if (jn < 0 || (jn > 0 && (ja < 0 || ja > m->njnt || jn > m->njnt - ja))) BAD("body_jntadr/num[%d] = %d/%d", b, ja, jn)
      if (dn < 0 || (dn > 0 && (da < 0 || da > m->nv || dn > m->nv - da))) BAD("body_dofadr/num[%d] = %d/%d", b, da, dn)
    }
    for (int j = 0; j < m->njnt; ++j) {
      const int ty = m->jnt_type[j], nqj = ty == MYO_JNT_FREE ? 7 : (ty == MYO_JNT_BALL ? 4 : 1), nvj = ty == MYO_JNT_FREE ? 6 : (ty == MYO_JNT_BALL ? 3 : 1);
      if (ty < 0 || ty > 3) BAD("jnt_type[%d] = %d", j, ty)
      if (m->jnt_qposadr[j] < 0 || m->jnt_qposadr[j] > m->nq - nqj) BAD("jnt_qposadr[%d] = %d", j, m->jnt_qposadr[j])
      if (m->jnt_dofadr[j] < 0 || m->jnt_dofadr[j] > m->nv - nvj) BAD("jnt_dofadr[%d] = %d", j, m->jnt_dofadr[j])
      if (m->jnt_bodyid[j] < 0 || m->jnt_bodyid[j] >= m->nbody) BAD("jnt_bodyid[%d] = %d", j, m->jnt_bodyid[j])
    }
    for (int d = 0; d < m->nv; ++d) {
      if (m->dof_bodyid[d] < 0 || m->dof_bodyid[d] >= m->nbody) BAD("dof_bodyid[%d] = %d", d, m->dof_bodyid[d])
      if (m->dof_jntid[d] < 0 || m->dof_jntid[d] >= m->njnt) BAD("dof_jntid[%d] = %d", d, m->dof_jntid[d])
      if (m->dof_parentid[d] < -1 || m->dof_parentid[d] >= d) BAD("dof_parentid[%d] = %d", d, m->dof_parentid[d])
    }
    for (int g = 0; g < m->ngeom; ++g) {
      if (m->geom_bodyid[g] < 0 || m->geom_bodyid[g] >= m->nbody) BAD("geom_bodyid[%d] = %d", g, m->geom_bodyid[g])
      if (m->geom_type[g] < 0 || m->geom_type[g] > MYO_GEOM_MESH) BAD("geom_type[%d] = %d", g, m->geom_type[g])
    }
    for (int k = 0; k < m->nsite; ++k) if (m->site_bodyid[k] < 0 || m->site_bodyid[k] >= m->nbody) BAD("site_bodyid[%d] = %d", k, m->site_bodyid[k])
    for (int t = 0; t < m->ntendon; ++t) {
      const int a = m->tendon_adr[t], c = m->tendon_num[t];
      if (a < 0 || c < 0 || a > m->nwrap || c > m->nwrap - a) BAD("tendon_adr/num[%d] = %d/%d", t, a, c)
    }
    for (int w = 0; w < m->nwrap; ++w) {
      const int ty = m->wrap_type[w], id = m->wrap_objid[w];
      if (ty == MYO_WRAP_SITE && (id < 0 || id >= m->nsite)) BAD("wrap_objid[%d] = %d (site)", w, id)
      if (ty == MYO_WRAP_SPHERE || ty == MYO_WRAP_CYLINDER) {
        if (id < 0 || id >= m->ngeom) BAD("wrap_objid[%d] = %d (geom)", w, id)
        if (m->wrap_prm[w] >= 0 && !(m->wrap_prm[w] < (double)m->nsite)) BAD("wrap_prm[%d]: side site out of range", w)
      }
    }
    for (int i = 0; i < m->nu; ++i) if (trntype[i] == MYO_TRN_TENDON && (trnid[2 * i] < 0 || trnid[2 * i] >= m->ntendon)) BAD("actuator_trnid[%d] = %d", i, trnid[2 * i])
    for (size_t p = 0; p < m->pair_geom1.size(); ++p)
      if (m->pair_geom1[p] < 0 || m->pair_geom1[p] >= m->ngeom || m->pair_geom2[p] < 0 || m->pair_geom2[p] >= m->ngeom) BAD("collision pair %zu names a geom out of range", p)
    if (!(m->timestep > 0) || !std::isfinite(m->timestep) || m->iterations < 0) BAD("opt.timestep / opt.iterations")
#undef BAD
  }
  {   // pairs of the primitive narrow phases first, the extended ones (csrc/myo_physics.h:collide_pair_ext) after them
    auto is_std = [&](int p) {
      const int t1 = m->geom_type[m->pair_geom1[p]], t2 = m->geom_type[m->pair_geom2[p]];
      return (t1 == MYO_GEOM_PLANE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE)) || (t1 == MYO_GEOM_SPHERE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE || t2 == MYO_GEOM_BOX)) ||
             (t1 == MYO_GEOM_CAPSULE && t2 == MYO_GEOM_CAPSULE);
    };
    int k = 0;
    while (k < m->npair && is_std(k)) k++;
    m->npair_std = k;
    for (; k < m->npair; ++k)
      if (is_std(k)) { int rc = fail(MYO_E_ARG, "corrupt model: collision pairs are not ordered (primitive pairs first)"); delete m; return rc; }
  }
  // ---- capacity / feature checks
#define LIM(cond, what) if (cond) { int rc = fail(MYO_E_UNSUPPORTED, "model exceeds stepper capacity: %s", what); delete m; return rc; }
  LIM(m->nbody > MYO_NB_MAX, "nbody") LIM(m->njnt > MYO_NJ_MAX, "njnt") LIM(m->nv > MYO_NV_MAX, "nv")
  LIM(m->nq > MYO_NQ_MAX, "nq") LIM(m->ntendon > MYO_NT_MAX, "ntendon") LIM(m->nu > MYO_NU_MAX, "nu")
  LIM(m->nbody > 64 || m->njnt > 64 || m->ntendon > 64 || m->nu > 64, "more than 64 bodies/joints/tendons/actuators")
  for (int i = 0; i < m->nu; ++i) LIM(trntype[i] != MYO_TRN_TENDON, "only tendon transmissions are supported")
  for (int j = 0; j < m->njnt; ++j) LIM(m->jnt_type[j] == MYO_JNT_BALL, "ball joints")
  // ---- derived tables
  const int nb = m->nbody, nv = m->nv;
  m->body_depth.assign(nb, 0);
  m->maxdepth = 0;
  for (int b = 1; b < nb; ++b) { m->body_depth[b] = m->body_depth[m->body_parentid[b]] + 1; if (m->body_depth[b] > m->maxdepth) m->maxdepth = m->body_depth[b]; }
  m->dof_rootbody.resize(nv);
  for (int d = 0; d < nv; ++d) m->dof_rootbody[d] = m->body_rootid[m->dof_bodyid[d]];
  m->body_dofmask.assign(nb, 0ull);
  for (int b = 1; b < nb; ++b) {
    unsigned long long mk = m->body_dofmask[m->body_parentid[b]];
    for (int k = 0; k < m->body_dofnum[b]; ++k) mk |= 1ull << (m->body_dofadr[b] + k);
    m->body_dofmask[b] = mk;
  }
  m->body_submask.assign(nb, 0ull);
  for (int b = nb - 1; b >= 1; --b) {
    m->body_submask[b] |= 1ull << b;
    if (m->body_parentid[b] > 0) m->body_submask[m->body_parentid[b]] |= m->body_submask[b];
  }
  m->dof_prevmask.assign(nv, 0ull);
  for (int d = 0; d < nv; ++d) {
    const int j = m->dof_jntid[d], b = m->dof_bodyid[d];
    unsigned long long below = (d == 0) ? 0ull : ((1ull << d) - 1ull);
    if (m->jnt_type[j] == MYO_JNT_FREE) {
      const int da = m->jnt_dofadr[j];
      if (d < da + 3) m->dof_prevmask[d] = 1ull << 63;  // translational: cdof_dot = 0
      else m->dof_prevmask[d] = (m->body_dofmask[b] & ((da == 0) ? 0ull : ((1ull << da) - 1ull))) | (7ull << da);
    } else m->dof_prevmask[d] = m->body_dofmask[b] & below;
  }
  // tree-sparse M (MuJoCo's dof_Madr order: row i holds (i,i),(i,parent),(i,grandparent),...)
  m->dof_Madr.resize(nv);
  m->M_i.clear(); m->M_j.clear();
  for (int i = 0; i < nv; ++i) {
    m->dof_Madr[i] = (int)m->M_i.size();
    for (int j = i; j >= 0; j = m->dof_parentid[j]) { m->M_i.push_back(i); m->M_j.push_back(j); }
  }
  m->nM = (int)m->M_i.size();
  // packed entry table: row | column << 8 | body of the row dof << 16 (one load per entry of the sparse M)
  m->M_pk.assign(MYO_NM_MAX, 0);
  for (int e = 0; e < m->nM && e < MYO_NM_MAX; ++e)
    m->M_pk[e] = m->M_i[e] | (m->M_j[e] << 8) | (m->dof_bodyid[m->M_i[e]] << 16);
  LIM(m->nM > MYO_NM_MAX, "nM")
  {
    std::vector<std::vector<std::pair<int, int>>> rows(nv);
    for (int e = 0; e < m->nM; ++e) {
      rows[m->M_i[e]].push_back({m->M_j[e], e});
      if (m->M_i[e] != m->M_j[e]) rows[m->M_j[e]].push_back({m->M_i[e], e});
    }
    m->mv_adr.assign(nv + 1, 0);
    for (int i = 0; i < nv; ++i) {
      m->mv_adr[i] = (int)m->mv_col.size();
      for (auto& pr : rows[i]) { m->mv_col.push_back(pr.first); m->mv_e.push_back(pr.second); }
    }
    m->mv_adr[nv] = (int)m->mv_col.size();
    // the same rows packed for one wide load per dof: 16-bit entries (qM index << 6 | column), two per word
    m->mv_pack.assign((size_t)nv * (MYO_MV_ROW / 2), 0);
    m->mv_len.assign(nv, 0);
    for (int i = 0; i < nv; ++i) {
      const int len = m->mv_adr[i + 1] - m->mv_adr[i];
      LIM(len > MYO_MV_ROW, "a row of the inertia matrix has more than MYO_MV_ROW non-zeros")
      m->mv_len[i] = len;
      for (int k = 0; k < len; ++k) {
        const unsigned ent = ((unsigned)m->mv_e[m->mv_adr[i] + k] << 6) | (unsigned)m->mv_col[m->mv_adr[i] + k];
        unsigned& w = reinterpret_cast<unsigned&>(m->mv_pack[(size_t)i * (MYO_MV_ROW / 2) + k / 2]);
        w |= (k & 1) ? (ent << 16) : ent;
      }
    }
  }
  // tendons: dofs each one can move; side sites
  m->tendon_dofmask.assign(m->ntendon, 0ull);
  m->wrap_side.assign(m->nwrap, -1);
  for (int t = 0; t < m->ntendon; ++t) {
    unsigned long long mk = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; ++w) {
      const int ty = m->wrap_type[w];
      if (ty == MYO_WRAP_SITE) mk |= m->body_dofmask[m->site_bodyid[m->wrap_objid[w]]];
      else if (ty == MYO_WRAP_SPHERE || ty == MYO_WRAP_CYLINDER) {
        mk |= m->body_dofmask[m->geom_bodyid[m->wrap_objid[w]]];
        m->wrap_side[w] = m->wrap_prm[w] >= 0 ? (int)lround(m->wrap_prm[w]) : -1;
      } else if (ty != MYO_WRAP_PULLEY) LIM(true, "fixed (joint) tendons")
    }
    m->tendon_dofmask[t] = mk;
    int cnt = 0;
    for (unsigned long long x = mk; x; x &= x - 1) cnt++;
    LIM(cnt > MYO_TJ_MAX, "a tendon moves more than MYO_TJ_MAX dofs")
  }
  // resolved wrap records: everything the tendon stage needs about wrap object w behind ONE level of
  // indexing (the stage is bound by dependent table loads otherwise: type -> objid -> bodyid -> pos):
  //   wr_i[8w..]  = type, body (site / geom body), geom id (-1), side-site body (-1), root body of
  //                 `body`, root body of the side-site body, 0, 0
  //   wr_p[4w..]  = local position (site_pos / geom_pos), pulley divisor
  //   wr_m[12w..] = geom_mat (9), side-site local position (3)
  //   wr_mask[w]  = body_dofmask[body]
  m->wr_i.assign(8 * (size_t)m->nwrap, 0);
  m->wr_p.assign(4 * (size_t)m->nwrap, 0.0);
  m->wr_m.assign(12 * (size_t)m->nwrap, 0.0);
  m->wr_mask.assign(m->nwrap, 0ull);
  for (int w = 0; w < m->nwrap; ++w) {
    const int ty = m->wrap_type[w], id = m->wrap_objid[w];
    int* I = &m->wr_i[8 * (size_t)w];
    I[0] = ty; I[1] = -1; I[2] = -1; I[3] = -1; I[4] = 0; I[5] = 0;
    if (ty == MYO_WRAP_SITE) {
      I[1] = m->site_bodyid[id];
      for (int k = 0; k < 3; ++k) m->wr_p[4 * (size_t)w + k] = m->site_pos[3 * id + k];
    } else if (ty == MYO_WRAP_SPHERE || ty == MYO_WRAP_CYLINDER) {
      I[1] = m->geom_bodyid[id]; I[2] = id;
      for (int k = 0; k < 3; ++k) m->wr_p[4 * (size_t)w + k] = m->geom_pos[3 * id + k];
      double gm[9];
      quat2mat_h(&geom_quat[4 * id], gm);
      for (int k = 0; k < 9; ++k) m->wr_m[12 * (size_t)w + k] = gm[k];
      const int sid = m->wrap_side[w];
      if (sid >= 0) {
        I[3] = m->site_bodyid[sid]; I[5] = m->body_rootid[I[3]];
        for (int k = 0; k < 3; ++k) m->wr_m[12 * (size_t)w + 9 + k] = m->site_pos[3 * sid + k];
      }
    } else if (ty == MYO_WRAP_PULLEY) m->wr_p[4 * (size_t)w + 3] = m->wrap_prm[w];
    if (I[1] >= 0) { I[4] = m->body_rootid[I[1]]; m->wr_mask[w] = m->body_dofmask[I[1]]; }
  }
  // geom wraps, enumerated: gw_elem[k] = path element of the k-th sphere/cylinder wrap; wr_i[8w+6] = k
  m->gw_elem.clear();
  for (int w = 0; w < m->nwrap; ++w) {
    m->wr_i[8 * (size_t)w + 6] = -1;
    if (m->wrap_type[w] == MYO_WRAP_SPHERE || m->wrap_type[w] == MYO_WRAP_CYLINDER) {
      m->wr_i[8 * (size_t)w + 6] = (int)m->gw_elem.size();
      m->gw_elem.push_back(w);
    }
  }
  m->ngw = (int)m->gw_elem.size();
  if (m->gw_elem.empty()) m->gw_elem.push_back(0);
  // path elements, enumerated (the walk mj_tendon makes along each tendon, resolved once): element e runs from the site
  // te_i[4e] to the site te_i[4e+1], around the wrap geom te_i[4e+2] (-1: straight), belongs to tendon te_i[4e+3] and
  // counts with 1 / te_div[e] (the last pulley before it).  The tendon stage gives each element its own lane.
  m->te_i.clear(); m->te_div.clear();
  m->tendon_eadr.assign(m->ntendon, 0); m->tendon_enum.assign(m->ntendon, 0);
  for (int t = 0; t < m->ntendon; ++t) {
    const int adr = m->tendon_adr[t], num = m->tendon_num[t];
    double divisor = 1.0;
    m->tendon_eadr[t] = (int)m->te_div.size();
    for (int j = 0; j < num - 1;) {
      const int ty0 = m->wrap_type[adr + j], ty1 = m->wrap_type[adr + j + 1];
      if (ty0 == MYO_WRAP_PULLEY || ty1 == MYO_WRAP_PULLEY) {
        if (ty0 == MYO_WRAP_PULLEY) divisor = m->wrap_prm[adr + j];
        j++;
        continue;
      }
      const int is_geom = (ty1 == MYO_WRAP_SPHERE || ty1 == MYO_WRAP_CYLINDER);
      LIM(is_geom && j + 2 >= num, "a tendon path ends on a wrap geom")
      const int end = j + (is_geom ? 2 : 1);
      m->te_i.push_back(adr + j); m->te_i.push_back(adr + end); m->te_i.push_back(is_geom ? adr + j + 1 : -1); m->te_i.push_back(t);
      m->te_div.push_back(divisor);
      j = end;
    }
    m->tendon_enum[t] = (int)m->te_div.size() - m->tendon_eadr[t];
  }
  m->nte = (int)m->te_div.size();
  // element lengths are staged (HP) in the part of H that is free during the tendon stage (behind cinert)
  LIM((size_t)m->nte * sizeof(double) > (size_t)(MYO_H_SIZE - MYO_NB_MAX * 10) * sizeof(float), "tendon path elements (length staging)")
  if (m->te_div.empty()) { m->te_i.assign(4, 0); m->te_div.push_back(1.0); }
  // staging area of the tendon stage.  Mixed stepper: T path points in con[], then (8-byte aligned) 7 HP wrap results per geom wrap,
  // running on through the limit-row, efc_* and solver vectors up to rk.  fp64 stepper: the wrap results only, up to the solver
  // vectors (where its body poses live during the position stage)
  typedef Scratch<double, MYO_NCON_F64> ScratchD;
  typedef Scratch<double, MYO_NCON_BIG> ScratchDB;
  // (... and not beyond efc_jv, where that stepper accumulates the tendon moment arms meanwhile)
  LIM(7 * (size_t)m->ngw * sizeof(double) > offsetof(ScratchD, efc_jv) - offsetof(ScratchD, con) ||
      m->ngw > MYO_BIGWS_GW ||                     /* (the 48-slot fp64 scratch: in the big workspace, TaskDev::big_ws) */
      ((3 * (size_t)m->nwrap * sizeof(float) + 7) & ~(size_t)7) + 7 * (size_t)m->ngw * sizeof(double) >
          offsetof(Scratch<float>, rk) - offsetof(Scratch<float>, con), "tendon path elements / wrap geoms (staging area of the tendon stage)")
  m->actuator_tendon.resize(m->nu);
  for (int i = 0; i < m->nu; ++i) m->actuator_tendon[i] = trnid[2 * i];
  // qfrc_actuator gather, dof-major: for dof d the (ten_J offset << 6 | actuator) pairs of every
  // actuator whose tendon moves d, in actuator order; 16-bit entries, two per word (one wide load per dof)
  m->aq_pack.assign((size_t)nv * (MYO_AQ_ROW / 2), 0);
  m->aq_len.assign(nv, 0);
  for (int d = 0; d < nv; ++d) {
    int len = 0;
    for (int i = 0; i < m->nu; ++i) {
      const int t = m->actuator_tendon[i];
      const unsigned long long mk = m->tendon_dofmask[t];
      if (!((mk >> d) & 1ull)) continue;
      LIM(len >= MYO_AQ_ROW, "more than MYO_AQ_ROW actuators act on one dof")
      int slot = 0;
      for (int b = 0; b < d; ++b) slot += (int)((mk >> b) & 1ull);
      const unsigned ent = ((unsigned)(t * MYO_TJ_MAX + slot) << 6) | (unsigned)i;
      unsigned& w = reinterpret_cast<unsigned&>(m->aq_pack[(size_t)d * (MYO_AQ_ROW / 2) + len / 2]);
      w |= (len & 1) ? (ent << 16) : ent;
      len++;
    }
    m->aq_len[d] = len;
  }
  // actuator-major copies for the moment-transpose gather, zero-padded to MYO_NU_MAX so the loop
  // runs in unconditional groups of 8 (scalar loads merge into s_load_dwordx8/x16)
  m->act_dofmask.assign(MYO_NU_MAX, 0ull);
  m->act_tj.assign(MYO_NU_MAX, 0);
  m->act_gear0.assign(MYO_NU_MAX, 0.0);
  m->act_sd.assign((size_t)MYO_NU_MAX * MYO_TJ_MAX, -1);      // dof of slot k of actuator i's tendon (-1: the tendon moves fewer dofs)
  for (int i = 0; i < m->nu; ++i) {
    m->act_dofmask[i] = m->tendon_dofmask[m->actuator_tendon[i]];
    m->act_tj[i] = m->actuator_tendon[i] * MYO_TJ_MAX;
    m->act_gear0[i] = m->actuator_gear[6 * i];
    { int k = 0; for (int d = 0; d < nv && k < MYO_TJ_MAX; ++d) if ((m->act_dofmask[i] >> d) & 1ull) m->act_sd[(size_t)i * MYO_TJ_MAX + k++] = d; }
  }
  // muscle constants of every actuator (fwd_actuation: what mju_muscleGain / mju_muscleBias / mju_muscleDynamics derive from the
  // parameters alone, with every constant denominator turned into a reciprocal): MYO_ACT_PRE doubles per actuator, see myo_physics.h
  m->act_pre.assign((size_t)MYO_NU_MAX * MYO_ACT_PRE, 0.0);
  for (int i = 0; i < m->nu; ++i) {
    double* P = &m->act_pre[(size_t)i * MYO_ACT_PRE];
    const double* gp = &m->actuator_gainprm[10 * (size_t)i];
    const double* bp = &m->actuator_biasprm[10 * (size_t)i];
    const double* dp = &m->actuator_dynprm[10 * (size_t)i];
    const double lr0 = m->actuator_lengthrange[2 * (size_t)i], lr1 = m->actuator_lengthrange[2 * (size_t)i + 1], acc0 = m->actuator_acc0[i];
    const double tiny = 1e-15;
    {   // gain
      const double force = gp[2] < 0 ? gp[3] / std::max(tiny, acc0) : gp[2];
      const double L0 = (lr1 - lr0) / std::max(tiny, gp[1] - gp[0]);
      const double lmin = gp[4], lmax = gp[5], a = 0.5 * (lmin + 1), b = 0.5 * (1 + lmax), y = gp[8] - 1;
      P[0] = force; P[1] = gp[0]; P[2] = 1.0 / std::max(tiny, L0); P[3] = lr0; P[4] = 1.0 / std::max(tiny, L0 * gp[6]);
      P[5] = lmin; P[6] = lmax; P[7] = a; P[8] = b;
      P[9] = 1.0 / std::max(tiny, a - lmin); P[10] = 1.0 / std::max(tiny, 1 - a); P[11] = 1.0 / std::max(tiny, b - 1); P[12] = 1.0 / std::max(tiny, lmax - b);
      P[13] = y; P[14] = 1.0 / std::max(tiny, y); P[15] = gp[8];
    }
    {   // bias
      const double force = bp[2] < 0 ? bp[3] / std::max(tiny, acc0) : bp[2];
      const double L0 = (lr1 - lr0) / std::max(tiny, bp[1] - bp[0]);
      const double b = 0.5 * (1 + bp[5]);
      P[16] = force; P[17] = bp[0]; P[18] = 1.0 / std::max(tiny, L0); P[19] = b; P[20] = 1.0 / std::max(tiny, b - 1); P[21] = bp[7];
    }
    // activation dynamics: 1 / tau_deact when tau = tau_deact / (0.5 + 1.5 act) can never reach the floor (act in [0, 1]); else 0: the formula as written
    P[22] = dp[1] / 2.0 >= tiny ? 1.0 / dp[1] : 0.0;
  }
  // per-lane records (myo_model_dev.h: bk_i / bk_f / jk_i / dk_i / dof_spr / dof_submask): what a body / joint / dof lane reads in a
  // stage, behind ONE index instead of body -> joint -> qpos chains of dependent vector loads
  {
    const int njnt = m->njnt;
    m->bk_i.assign((size_t)MYO_BK_I * (nb > 0 ? nb : 1), 0);
    m->bk_f.assign((size_t)MYO_BK_F * (nb > 0 ? nb : 1), 0.0);
    for (int b = 0; b < nb; ++b) {
      int* I = &m->bk_i[(size_t)MYO_BK_I * b];
      double* F = &m->bk_f[(size_t)MYO_BK_F * b];
      const int jn = m->body_jntnum[b], ja = m->body_jntadr[b];
      const bool is_free = jn == 1 && m->jnt_type[ja] == MYO_JNT_FREE;
      LIM(jn > MYO_BK_NJ, "more than three joints on one body")
      I[0] = m->body_depth[b]; I[1] = m->body_parentid[b]; I[2] = jn; I[3] = ja; I[4] = is_free ? 1 : 0;
      for (int k = 0; k < 3; ++k) F[k] = m->body_pos[3 * (size_t)b + k];
      for (int k = 0; k < 4; ++k) F[3 + k] = m->body_quat[4 * (size_t)b + k];
      for (int k = 0; k < jn && k < MYO_BK_NJ; ++k) {
        const int j = ja + k, qa = m->jnt_qposadr[j];
        I[5 + k] = qa; I[8 + k] = m->jnt_type[j];
        for (int e = 0; e < 3; ++e) { F[7 + 7 * k + e] = m->jnt_pos[3 * (size_t)j + e]; F[10 + 7 * k + e] = m->jnt_axis[3 * (size_t)j + e]; }
        F[13 + 7 * k] = m->qpos0[qa];
      }
    }
    m->jk_i.assign((size_t)4 * (njnt > 0 ? njnt : 1), 0);
    for (int j = 0; j < njnt; ++j) {
      int* I = &m->jk_i[(size_t)4 * j];
      I[0] = m->jnt_bodyid[j]; I[1] = m->jnt_dofadr[j]; I[2] = m->jnt_type[j]; I[3] = m->body_rootid[m->jnt_bodyid[j]];
    }
    m->dk_i.assign((size_t)2 * (nv > 0 ? nv : 1), 0);
    m->dof_spr.assign((size_t)2 * (nv > 0 ? nv : 1), 0.0);
    m->dof_submask.assign(nv > 0 ? nv : 1, 0ull);
    for (int d = 0; d < nv; ++d) {
      const int j = m->dof_jntid[d], qa = m->jnt_qposadr[j];
      m->dk_i[(size_t)2 * d] = qa;
      const bool spring = m->jnt_type[j] != MYO_JNT_FREE && m->jnt_stiffness[j] != 0;
      m->dof_spr[(size_t)2 * d] = spring ? m->jnt_stiffness[j] : 0.0;
      m->dof_spr[(size_t)2 * d + 1] = m->qpos_spring[qa];
      m->dof_submask[d] = m->body_submask[m->dof_bodyid[d]];
    }
  }
  for (int b = 0; b < nb; ++b) {
    int cnt = 0;
    for (unsigned long long x = m->body_dofmask[b]; x; x &= x - 1) cnt++;
    LIM(2 * cnt > MYO_CS_MAX + 4 && false, "contact support")
  }
  // per-pair contact records: everything mj_contactParam / the constraint build derive from the two geoms
  // alone is resolved here (the device stage was a chain of dependent table loads per contact):
  //   pc_i[8p..]  = body1, body2, root body 1, root body 2, nsup, box-box candidate (0 none; 1 + v: vertex v of geom 1 against
  //                 geom 2, 9 + v: vertex v of geom 2 against geom 1, 17: the edge-edge candidate), 0, friction selector (0 max, 1 geom1, 2 geom2)
  //   pc_sup[4p..] = the dofs either body can move (<= 16 bytes, ascending)
  //   pc_f[16p..] = margin, margin - gap, (K, B) and (d0, d1, 1 / width, mid, power) of the mixed solref[2] / solimp[5] (sol_precompute), friction1[3], friction2[3], invweight sum
  //   pc_mask[2p..] = ancestor-dof masks of the two bodies
  {
    const int np = m->npair > 0 ? m->npair : 1;
    m->pc_i.assign(8 * (size_t)np, 0); m->pc_sup.assign(4 * (size_t)np, 0);
    m->pc_f.assign(16 * (size_t)np, 0.0); m->pc_mask.assign(2 * (size_t)np, 0ull);
    m->pair_mg.assign(2 * (size_t)np, 0.0);     // margin, margin - gap per pair row (HP copy on the device: the general narrow-phase path reads them)
  }
  m->any_rot = 0; m->any_gen = 0;
  for (int p = 0; p < m->npair; ++p) {
    const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    const unsigned long long m1 = m->body_dofmask[b1], m2 = m->body_dofmask[b2];
    unsigned long long mk = m1 | m2;
    int cnt = 0;
    for (unsigned long long x = mk; x; x &= x - 1) cnt++;
    LIM(cnt > MYO_CS_MAX, "a contact pair moves more than MYO_CS_MAX dofs")
    int* I = &m->pc_i[8 * (size_t)p];
    I[0] = b1; I[1] = b2; I[2] = m->body_rootid[b1]; I[3] = m->body_rootid[b2]; I[4] = cnt; I[5] = pair_sub[p];
    unsigned char* sup = reinterpret_cast<unsigned char*>(&m->pc_sup[4 * (size_t)p]);
    // (bit 6 / bit 7 of an entry: ancestor dof of body 1 / body 2 — ContactRec::sup)
    { int ns = 0; for (int d = 0; d < 64 && ns < MYO_CS_MAX; ++d) if ((mk >> d) & 1ull) sup[ns++] = (unsigned char)(d | (int)((m1 >> d) & 1ull) << 6 | (int)((m2 >> d) & 1ull) << 7); }
    m->pc_mask[2 * (size_t)p] = m1; m->pc_mask[2 * (size_t)p + 1] = m2;
    const int pr1 = m->geom_priority[g1], pr2 = m->geom_priority[g2];
    double mix;
    if (pr1 != pr2) mix = pr1 > pr2 ? 1.0 : 0.0;
    else {
      const double x1 = m->geom_solmix[g1], x2 = m->geom_solmix[g2];
      const double tiny = 1e-15;      // MYO_MINVAL
      if (x1 >= tiny && x2 >= tiny) mix = x1 / (x1 + x2);
      else if (x1 < tiny && x2 < tiny) mix = 0.5;
      else mix = x1 < tiny ? 0.0 : 1.0;
    }
    double* F = &m->pc_f[16 * (size_t)p];
    const double margin = std::max(m->geom_margin[g1], m->geom_margin[g2]);
    F[0] = margin; F[1] = margin - std::max(m->geom_gap[g1], m->geom_gap[g2]);
    const double *r1 = &m->geom_solref[2 * g1], *r2 = &m->geom_solref[2 * g2];
    for (int e = 0; e < 2; ++e) F[2 + e] = (r1[0] > 0 && r2[0] > 0) ? mix * r1[e] + (1 - mix) * r2[e] : std::min(r1[e], r2[e]);
    for (int e = 0; e < 5; ++e) F[4 + e] = mix * m->geom_solimp[5 * g1 + e] + (1 - mix) * m->geom_solimp[5 * g2 + e];
    for (int e = 0; e < 3; ++e) { F[9 + e] = m->geom_friction[3 * g1 + e]; F[12 + e] = m->geom_friction[3 * g2 + e]; }
    F[15] = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    I[7] = (pr1 == pr2) ? 0 : (pr1 > pr2 ? 1 : 2);
    // condim of the contact (mj_contactParam): the higher-priority geom's, the larger of the two at equal priority
    const int d1 = geom_condim[g1], d2 = geom_condim[g2];
    I[6] = (pr1 == pr2) ? std::max(d1, d2) : (pr1 > pr2 ? d1 : d2);
    double mg0 = margin, mg1 = F[1];
    if (pair_xp[p] >= 0) {                    // an explicit <pair>: its own margin / gap / solref / solimp / friction / condim, nothing mixed, no per-env friction
      const int x = pair_xp[p];
      mg0 = xp_margin[x]; mg1 = xp_margin[x] - xp_gap[x];
      F[0] = mg0; F[1] = mg1;
      for (int e = 0; e < 2; ++e) F[2 + e] = xp_solref[2 * x + e];
      for (int e = 0; e < 5; ++e) F[4 + e] = xp_solimp[5 * x + e];
      for (int e = 0; e < 3; ++e) { F[9 + e] = xp_friction[3 * x + e]; F[12 + e] = xp_friction[3 * x + e]; }
      I[6] = xp_dim[x] | 0x100;
      I[7] = 0;
      m->any_gen = 1;                         // (the general path reads the pair's own margin and skips the per-env friction patch)
    }
    m->pair_mg[2 * (size_t)p] = mg0; m->pair_mg[2 * (size_t)p + 1] = mg1;
    sol_precompute(F + 2, F + 4, m->timestep, m->disableflags);      // pc_f[2..8]: (K, B), (d0, d1, 1 / width, mid, power) — what sol_param reads
    LIM((I[6] & 255) != 1 && (I[6] & 255) != 3 && (I[6] & 255) != 4 && (I[6] & 255) != 6, "contact dimension (condim) other than 1, 3, 4, 6")
    if ((I[6] & 255) > 3) m->any_rot = 1;
    if ((I[6] & 255) != 3) m->any_gen = 1;
  }
  m->body_imat.resize(9 * nb);
  for (int b = 0; b < nb; ++b) quat2mat_h(&body_iquat[4 * b], &m->body_imat[9 * b]);
  m->geom_mat.resize(9 * (size_t)m->ngeom);
  for (int g = 0; g < m->ngeom; ++g) quat2mat_h(&geom_quat[4 * g], &m->geom_mat[9 * g]);
  // trailing dofs whose row of M is diagonal by construction: a free joint on a leaf body whose
  // inertial frame coincides with the body frame (M = diag(m,m,m,Ixx,Iyy,Izz)); nlead = first of them
  m->nlead = nv;
  for (int j = m->njnt - 1; j >= 0; --j) {
    const int b = m->jnt_bodyid[j];
    bool leaf = true;
    for (int o = 1; o < nb; ++o) if (m->body_parentid[o] == b) leaf = false;
    const double* iq = &body_iquat[4 * b]; const double* ip = &m->body_ipos[3 * b];
    const bool aligned = fabs(fabs(iq[0]) - 1.0) < 1e-12 && fabs(ip[0]) + fabs(ip[1]) + fabs(ip[2]) < 1e-14;
    if (m->jnt_type[j] == MYO_JNT_FREE && leaf && aligned && m->body_jntnum[b] == 1 && m->jnt_dofadr[j] + 6 == m->nlead)
      m->nlead = m->jnt_dofadr[j];
    else break;
  }
  // the rows' solver parameters in the form sol_param reads them (in place: same names, same sizes)
  for (int j = 0; j < m->njnt; ++j) sol_precompute(&m->jnt_solref[2 * (size_t)j], &m->jnt_solimp[5 * (size_t)j], m->timestep, m->disableflags);
  for (int t2 = 0; t2 < m->ntendon; ++t2) {
    sol_precompute(&m->tendon_solref_lim[2 * (size_t)t2], &m->tendon_solimp_lim[5 * (size_t)t2], m->timestep, m->disableflags);
    sol_precompute(&m->tendon_solref_fri[2 * (size_t)t2], &m->tendon_solimp_fri[5 * (size_t)t2], m->timestep, m->disableflags);
  }
  for (int d = 0; d < nv; ++d) sol_precompute(&m->dof_solref[2 * (size_t)d], &m->dof_solimp[5 * (size_t)d], m->timestep, m->disableflags);
  build_ldl_tables(m);
  build_arrow_tables(m);
  m->any_floss = 0;
  {
    int nfr = 0;
    for (int d = 0; d < nv; ++d) if (m->dof_frictionloss[d] > 0) { m->any_floss = 1; nfr++; }
    for (int t = 0; t < m->ntendon; ++t) if (m->tendon_frictionloss[t] > 0) { m->any_floss = 1; nfr++; }
    LIM(nfr > MYO_NLIM_MAX / 2, "friction-loss rows (more than half of the limit-row capacity)")
    for (int d = 0; d < nv; ++d) LIM(m->dof_frictionloss[d] > 0 && m->jnt_type[m->dof_jntid[d]] == MYO_JNT_FREE, "friction loss on the dofs of a free joint")
  }
  m->any_damping = 0;
  for (int d = 0; d < nv; ++d) if (m->dof_damping[d] > 0) m->any_damping = 1;
  m->any_tendon_passive = 0;
  for (int t = 0; t < m->ntendon; ++t) if (m->tendon_stiffness[t] != 0 || m->tendon_damping[t] != 0) m->any_tendon_passive = 1;
#undef LIM
  *out = m;
  return MYO_OK;
}

extern "C" int myo_model_load_mjb(const char* path, int integrator, int unsupported_contacts, myo_model** out) {
  if (!path || !out) return fail(MYO_E_ARG, "myo_model_load_mjb: null argument");
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(MYO_E_ARG, "cannot open %s", path);
  try {                                   // the file is untrusted; no C++ exception crosses the C ABI
    std::vector<unsigned char> raw;
    unsigned char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, fh)) > 0) {
      raw.insert(raw.end(), buf, buf + k);
      if (raw.size() > ((size_t)1 << 31)) { fclose(fh); return fail(MYO_E_ARG, "%s: larger than 2 GiB", path); }
    }
    const int rd_err = ferror(fh);
    fclose(fh);
    if (rd_err) return fail(MYO_E_ARG, "%s: read error", path);
    myo_mjb::File f;
    std::string err;
    if (!myo_mjb::parse(raw.data(), raw.size(), f, err)) return fail(MYO_E_ARG, "%s: %s", path, err.c_str());
    std::vector<unsigned char> blob;
    int unsupported = 0;
    if (!myo_mjb::to_blob(f, integrator, unsupported_contacts, blob, err, &unsupported))
      return fail(unsupported ? MYO_E_UNSUPPORTED : MYO_E_ARG, "%s: %s", path, err.c_str());
    return myo_model_from_blob(blob.data(), blob.size(), out);
  } catch (const std::exception& e) {
    return fail(MYO_E_ARG, "%s: rejected (%s)", path, e.what());
  }
}
extern "C" void myo_model_destroy(myo_model* m) { delete m; }
extern "C" int myo_model_size(const myo_model* m, const char* n) {
  if (!m || !n) return -1;
#define S(x) if (!strcmp(n, #x)) return m->x;
  S(nq) S(nv) S(nu) S(na) S(nbody) S(njnt) S(ngeom) S(nsite) S(ntendon) S(nwrap) S(npair) S(nM) S(integrator) S(nlead) S(arrow_nf) S(ld_nfq) S(ld_nsq)
#undef S
  return -1;
}

// ------------------------------------------------------------------------------------------ batch
#define MYO_PARTS_MAX 8
struct StepPlan { int nparts; int k[MYO_PARTS_MAX + 1]; int wt; };   // wt: parts publish their record with write-through stores instead of an agent release fence      // part p = substeps [k[p], k[p+1]); nparts = 1: whole steps
struct myo_batch;
struct myo_batch {
  int n, device, dtype, nobs;
  int ncap;                    // contact capacity of the scratch: MYO_NCON_MAX, or MYO_NCON_BIG for models with extended pairs / a die
  myo_task_cfg cfg;
  TaskDev K;
  EnvRecordLayout L;
  DumpLayout D;
  double* rec;                 // dev [n, L.stride]
  std::vector<void*> allocs;   // every device allocation (model arrays, records)
  std::vector<double> geom_friction;   // host copy of the model's (nominal values of an object group)
  DevModel<double> Md;
  DevModel<float> Mf;
  int nq, nv, nu, na, nbody, nsite, ntendon, ngeom, integrator;
  unsigned char* bad_state;    // caller-owned dev uint8[n] or null (myo_batch_set_bad_state_buffer)
  // launch order of myo_batch_step (longest-predicted-first, see k_step_order): order[blockIdx] = env, cost[env] = smoothed duration
  // of the env's last steps, ticks[env] = duration of its last step (100 MHz ticks).  order = null: identity.
  int* order;
  float* cost;
  unsigned int* ticks;
  // env steps in parts (k_step): part_state[env] = 16 g + 2 q (parts < q of step g published) or + 1 (part q claimed, running);
  // step_gen[0] = g, advanced on the stream after every step; a workgroup that meets a state of another generation counts it in
  // K.health[0] (myo_batch_health).  plan = substep boundaries of the parts.
  int* part_state;
  int* step_gen;
  StepPlan plan;
  int timing;
  double ms_sum;
  int ms_cnt;
  int wrap_tune_in = 16;       // steps until the next census of the wraps (myo_batch_step: 16 steps after a reset of all envs, then every 256)
  int* wrap_cnt = nullptr;     // dev int[ngw]: engagement counts of the wrap census (k_wrap_census / k_wrap_reorder); null = one pass of wraps, nothing to order
  bool has_slot_ws = false;    // K.ctrl_ws is the device's shared wave-slot workspace (slot_workspace_acquire / _release)
  bool has_big_ws = false;     // ... and K.big_ws its block of the 48-slot fp64 scratch's records / wrap results
  MYO_BACKEND_BATCH_FIELDS     // what the backend keeps per batch (the HIP backend: its timing events)
};

template <typename T>
static int upload_model(const myo_model* m, DevModel<T>& D, std::vector<void*>& allocs) {
  D.nq = m->nq; D.nv = m->nv; D.nu = m->nu; D.na = m->na; D.nbody = m->nbody; D.njnt = m->njnt; D.ngeom = m->ngeom;
  D.nsite = m->nsite; D.ntendon = m->ntendon; D.nwrap = m->nwrap; D.npair = m->npair; D.nM = m->nM; D.maxdepth = m->maxdepth;
  D.integrator = m->integrator; D.iterations = m->iterations; D.disableflags = m->disableflags;
  D.any_damping = m->any_damping; D.any_tendon_passive = m->any_tendon_passive; D.nlead = m->nlead; D.ngw = m->ngw; D.nte = m->nte; D.npair_std = m->npair_std;
  D.ld_nfq = m->ld_nfq; D.ld_nsq = m->ld_nsq; D.arrow_nf = m->arrow_nf; D.arrow_pad = m->arrow_pad; D.any_rot = m->any_rot; D.any_gen = m->any_gen; D.any_floss = m->any_floss;
  D.h_timestep = m->timestep;
  D.timestep = (T)m->timestep; D.tolerance = (T)m->tolerance; D.impratio = (T)m->impratio;
  D.isqrt_impratio = (T)(1.0 / sqrt(m->impratio));
  for (int k = 0; k < 3; ++k) D.gravity[k] = (T)m->gravity[k];
  D.meaninertia = (T)m->meaninertia;
  int rc = 0;
#define X(n)                                                                            \
  {                                                                                     \
    void* p = nullptr;                                                                  \
    rc |= be_malloc(&p, m->n.size() * sizeof(int));                                     \
    if (!rc && !m->n.empty()) rc |= be_h2d(p, m->n.data(), m->n.size() * sizeof(int));  \
    allocs.push_back(p);                                                                \
    D.n.p = (const int*)p;                                                              \
  }
  MYO_MODEL_INT_ARRAYS(X)
#undef X
#define X(n)                                                                                               \
  {                                                                                                        \
    void* p = nullptr;                                                                                     \
    rc |= be_malloc(&p, m->n.size() * sizeof(unsigned long long));                                         \
    if (!rc && !m->n.empty()) rc |= be_h2d(p, m->n.data(), m->n.size() * sizeof(unsigned long long));      \
    allocs.push_back(p);                                                                                   \
    D.n.p = (const unsigned long long*)p;                                                                  \
  }
  MYO_MODEL_U64_ARRAYS(X)
#undef X
#define X(n)                                                                  \
  {                                                                           \
    std::vector<T> tmp(m->n.begin(), m->n.end());                             \
    void* p = nullptr;                                                        \
    rc |= be_malloc(&p, tmp.size() * sizeof(T));                              \
    if (!rc && !tmp.empty()) rc |= be_h2d(p, tmp.data(), tmp.size() * sizeof(T)); \
    allocs.push_back(p);                                                      \
    D.n.p = (const T*)p;                                                      \
  }
  MYO_MODEL_REAL_ARRAYS(X)
#undef X
  // HP tables: the fp64 stepper's ordinary tables; separate fp64 copies for the mixed stepper
#define X(n)                                                                                   \
  if (sizeof(T) == sizeof(double)) D.h_##n.p = (const double*)(const void*)D.n.p;              \
  else {                                                                                       \
    void* p = nullptr;                                                                         \
    rc |= be_malloc(&p, m->n.size() * sizeof(double));                                         \
    if (!rc && !m->n.empty()) rc |= be_h2d(p, m->n.data(), m->n.size() * sizeof(double));      \
    allocs.push_back(p);                                                                       \
    D.h_##n.p = (const double*)p;                                                              \
  }
  MYO_MODEL_HP_ARRAYS(X)
#undef X
  return rc;
}

static void make_taskdev(const myo_task_cfg* c, uint64_t seed, TaskDev& K) {
  memset(&K, 0, sizeof K);
  K.kind = MYO_TASK_NONE; K.frame_skip = 1; K.max_episode_steps = 1 << 30;
  K.obj1_sid = K.obj2_sid = K.target1_sid = K.target2_sid = -1;
  K.obj1_bid = K.obj2_bid = K.obj1_gid = K.obj2_gid = -1;
  K.objg_gid0 = K.objg_gidn = -1;
  K.seed = seed;
  if (!c) return;
  K.kind = c->kind; K.frame_skip = c->frame_skip; K.max_episode_steps = c->max_episode_steps; K.n_hand = c->n_hand;
  K.obj1_sid = c->obj1_sid; K.obj2_sid = c->obj2_sid; K.target1_sid = c->target1_sid; K.target2_sid = c->target2_sid;
  K.obj1_bid = c->obj1_bid; K.obj2_bid = c->obj2_bid; K.obj1_gid = c->obj1_gid; K.obj2_gid = c->obj2_gid;
  K.task_choice = c->task_choice; K.enable_rsi = c->enable_rsi; K.balls_overlap = c->balls_overlap;
  K.limit_init_angle_on = c->limit_init_angle_on; K.beta_init_angle_on = c->beta_init_angle_on;
  K.beta_ball_size_on = c->beta_ball_size_on; K.beta_ball_mass_on = c->beta_ball_mass_on;
  K.drop_th = c->drop_th; K.proximity_th = c->proximity_th;
  memcpy(K.center_pos, c->center_pos, sizeof K.center_pos); memcpy(K.weights, c->weights, sizeof K.weights);
  memcpy(K.goal_time_period, c->goal_time_period, 16); memcpy(K.goal_xrange, c->goal_xrange, 16);
  memcpy(K.goal_yrange, c->goal_yrange, 16);
  K.rsi_probability = c->rsi_probability; K.overlap_probability = c->overlap_probability;
  K.noise_palm = c->noise_palm; K.noise_fingers = c->noise_fingers; K.noise_balls = c->noise_balls;
  K.limit_init_angle = c->limit_init_angle;
  memcpy(K.beta_init_angle, c->beta_init_angle, 16); memcpy(K.beta_ball_size, c->beta_ball_size, 16);
  memcpy(K.beta_ball_mass, c->beta_ball_mass, 16); memcpy(K.obj_size_range, c->obj_size_range, 16);
  memcpy(K.obj_mass_range, c->obj_mass_range, 16); memcpy(K.obj_friction_change, c->obj_friction_change, 24);
  K.init_qpos0 = c->init_qpos0;
  if (c->kind == MYO_TASK_REORIENT) {
    // the Baoding per-ball overrides (mass / size / friction by body and geom id) must not fire: the die is the object GROUP
    K.ro_obj_bid = c->obj1_bid; K.objg_gid0 = c->obj1_gid; K.objg_gidn = c->obj2_gid;
    K.obj1_bid = K.obj2_bid = K.obj1_gid = K.obj2_gid = -1;
    memcpy(K.ro_weights, c->ro_weights, sizeof K.ro_weights); memcpy(K.ro_goal_pos, c->ro_goal_pos, 16);
    memcpy(K.ro_goal_rot, c->ro_goal_rot, 16); memcpy(K.ro_rot_choice, c->ro_rot_choice, sizeof K.ro_rot_choice);
    for (int k = 0; k < 3; ++k) K.ro_n_rot_choice[k] = c->ro_n_rot_choice[k];
    K.ro_obj_size_change = c->ro_obj_size_change; K.ro_pos_th = c->ro_pos_th; K.ro_rot_th = c->ro_rot_th;
    memcpy(K.ro_goal_init_pos, c->ro_goal_init_pos, 24); memcpy(K.ro_goal_obj_offset, c->ro_goal_obj_offset, 24);
  }
}
static_assert(MYO_TASK_REORIENT == MYO_TASK_REORIENT_K, "task kind ids");
static int task_nobs_host(const myo_task_cfg* c, int na) {
  return c->kind == MYO_TASK_REORIENT ? 2 * c->n_hand + 18 + na : c->n_hand + 24 + na;
}

extern "C" int myo_batch_create(const myo_model* m, const myo_task_cfg* cfg, int n_envs, int device, uint64_t seed,
                                int dtype, myo_batch** out) {
  if (!m || !out || n_envs <= 0) return fail(MYO_E_ARG, "bad arguments to myo_batch_create");
  if (dtype != MYO_F64 && dtype != MYO_F32) return fail(MYO_E_ARG, "dtype must be MYO_F64 or MYO_F32");
  if (cfg && cfg->kind != MYO_TASK_NONE) {
    if (cfg->kind != MYO_TASK_BAODING_P1 && cfg->kind != MYO_TASK_BAODING_P2 && cfg->kind != MYO_TASK_REORIENT)
      return fail(MYO_E_ARG, "unknown task kind");
    if (cfg->kind == MYO_TASK_REORIENT) {
      if (cfg->obj1_sid < 0 || cfg->obj1_sid >= m->nsite || cfg->target1_sid < 0 || cfg->target1_sid >= m->nsite ||
          cfg->obj1_bid <= 0 || cfg->obj1_bid >= m->nbody || cfg->obj1_gid < 0 || cfg->obj2_gid <= cfg->obj1_gid || cfg->obj2_gid > m->ngeom)
        return fail(MYO_E_ARG, "task ids out of range");
      if (cfg->obj2_gid - cfg->obj1_gid > MYO_OBJG_MAX) return fail(MYO_E_UNSUPPORTED, "the die has more than %d geoms", MYO_OBJG_MAX);
      if (cfg->n_hand + 7 != m->nq || m->nv < 6 || task_nobs_host(cfg, m->na) > MYO_OBS_MAX)
        return fail(MYO_E_UNSUPPORTED, "reorient task needs nq = n_hand + 7 (the die's free joint last) and an observation of <= %d numbers", MYO_OBS_MAX);
      for (int k = 0; k < 3; ++k)
        if (cfg->ro_n_rot_choice[k] < 0 || cfg->ro_n_rot_choice[k] > MYO_ROT_CHOICE_MAX) return fail(MYO_E_ARG, "goal_rot_x/y/z: at most %d ranges", MYO_ROT_CHOICE_MAX);
    } else {
    if (cfg->obj1_sid < 0 || cfg->obj1_sid >= m->nsite || cfg->obj2_sid < 0 || cfg->obj2_sid >= m->nsite ||
        cfg->target1_sid < 0 || cfg->target1_sid >= m->nsite || cfg->target2_sid < 0 || cfg->target2_sid >= m->nsite ||
        cfg->obj1_bid <= 0 || cfg->obj1_bid >= m->nbody || cfg->obj2_bid <= 0 || cfg->obj2_bid >= m->nbody ||
        cfg->obj1_gid < 0 || cfg->obj1_gid >= m->ngeom || cfg->obj2_gid < 0 || cfg->obj2_gid >= m->ngeom)
      return fail(MYO_E_ARG, "task ids out of range");
    if (cfg->n_hand + 14 != m->nq || m->nv < 12 || cfg->n_hand + 24 + m->na > MYO_OBS_MAX)
      return fail(MYO_E_UNSUPPORTED, "Baoding task needs nq = n_hand + 14 (two free balls last)");
    }
    if (cfg->frame_skip <= 0 || cfg->max_episode_steps <= 0) return fail(MYO_E_ARG, "frame_skip / max_episode_steps");
  }
  int rc = be_set_device(device);
  if (rc) return fail(MYO_E_DEVICE, "hipSetDevice(%d): %s", device, be_errstr(rc));
  myo_batch* b = new myo_batch();
  b->n = n_envs; b->device = device; b->dtype = dtype; b->bad_state = nullptr; b->timing = 0; b->ms_sum = 0; b->ms_cnt = 0;
  b->integrator = m->integrator;
  // contact capacity of the per-env scratch: the Baoding hand (39 candidate pairs, 11 contacts at most in the bench workload) keeps
  // the base; models with extended pairs (boxes, cylinders, ellipsoids) and the die task get the larger scratch (one workgroup
  // per CU less: 21.9 KB instead of 20.2 KB of LDS)
  // (a condim-4 / 6 contact takes two / three slots of the scratch: models that have such pairs get the big one too)
  b->ncap = (m->npair > m->npair_std || m->any_rot || (cfg && cfg->kind == MYO_TASK_REORIENT)) ? MYO_NCON_BIG : MYO_NCON_MAX;
  b->geom_friction = m->geom_friction;
  b->nq = m->nq; b->nv = m->nv; b->nu = m->nu; b->na = m->na; b->nbody = m->nbody; b->nsite = m->nsite; b->ntendon = m->ntendon; b->ngeom = m->ngeom;
  if (cfg) b->cfg = *cfg; else memset(&b->cfg, 0, sizeof b->cfg);
  make_taskdev(cfg && cfg->kind != MYO_TASK_NONE ? cfg : nullptr, seed, b->K);
  b->nobs = b->K.kind ? task_nobs_host(cfg, m->na) : 0;
  memset(&b->Md, 0, sizeof b->Md); memset(&b->Mf, 0, sizeof b->Mf);
  rc = (dtype == MYO_F64) ? upload_model<double>(m, b->Md, b->allocs) : upload_model<float>(m, b->Mf, b->allocs);
  EnvRecordLayout& L = b->L;
  L.nq = m->nq; L.nv = m->nv; L.na = m->na;
  int o = 0;
  L.off_qpos = o; o += m->nq; L.off_qvel = o; o += m->nv; L.off_act = o; o += m->na; L.off_warm = o; o += m->nv;
  L.off_time = o; o += 1; L.off_taskd = o; o += MYO_TASKD_N; L.off_balld = o; o += MYO_BALLD_N; L.off_misc = o; o += MYO_MISC_N;
  L.off_objfric = o; o += 3 * MYO_OBJG_MAX;
  // (store_env writes qpos | qvel | act | warm | time as one run: the order above is part of the kernels' contract)
  if (L.off_qvel != L.off_qpos + m->nq || L.off_act != L.off_qvel + m->nv || L.off_warm != L.off_act + m->na || L.off_time != L.off_warm + m->nv) { delete b; return fail(MYO_E_STATE, "record layout"); }
  L.stride = (o + 15) / 16 * 16;      // whole 128-byte lines per env: no line is shared by two workgroups
  DumpLayout& D = b->D;
  o = 0;
  D.ten_length = o; o += m->ntendon; D.ten_J = o; o += m->ntendon * m->nv; D.M = o; o += m->nv * m->nv;
  D.qfrc_bias = o; o += m->nv; D.qfrc_passive = o; o += m->nv; D.qfrc_actuator = o; o += m->nv;
  D.qacc_smooth = o; o += m->nv; D.qacc = o; o += m->nv; D.actuator_force = o; o += m->nu; D.act_dot = o; o += m->na;
  D.counts = o; o += 4; D.efc_aref = o; o += MYO_NLIM_MAX + 4 * MYO_NCON_BIG; D.efc_D = o; o += MYO_NLIM_MAX + 4 * MYO_NCON_BIG;
  D.site_xpos = o; o += 3 * m->nsite; D.subtree_com = o; o += 3 * m->nbody; D.xpos = o; o += 3 * m->nbody; D.total = o;
  // initial records: qpos0, zero velocity; model ball parameters; target sites at their model xy;
  // start angles per _setup (baoding.py:246-247 P1; :349-354 P2 decided per env at reset time here)
  std::vector<double> host((size_t)n_envs * L.stride, 0.0);
  for (int e = 0; e < n_envs; ++e) {
    double* r = &host[(size_t)e * L.stride];
    for (int i = 0; i < m->nq; ++i) r[L.off_qpos + i] = m->qpos0[i];
    double* td = r + L.off_taskd; double* bd = r + L.off_balld; double* mi = r + L.off_misc;
    td[0] = 3.0 * MYO_PI / 4.0; td[1] = -MYO_PI / 4.0; td[2] = 0.025; td[3] = 0.028; td[4] = 5.0;
    mi[0] = MYO_WHICH_CCW;
    if (b->K.kind == MYO_TASK_REORIENT) {
      // goal at its setup pose, identity orientation, nominal die; reset() draws the episode's values
      memset(td, 0, sizeof(double) * MYO_TASKD_N); mi[0] = 0;
      for (int k = 0; k < 3; ++k) td[k] = b->K.ro_goal_init_pos[k];
      td[3] = 1.0;
      for (int j = 0; j < 3 * (b->K.objg_gidn - b->K.objg_gid0); ++j) r[L.off_objfric + j] = m->geom_friction[3 * b->K.objg_gid0 + j];
    } else if (b->K.kind) {
      td[2] = 0.5 * (b->K.goal_xrange[0] + b->K.goal_xrange[1]); td[3] = 0.5 * (b->K.goal_yrange[0] + b->K.goal_yrange[1]);
      td[4] = 0.5 * (b->K.goal_time_period[0] + b->K.goal_time_period[1]);
      if (b->K.kind == MYO_TASK_BAODING_P2 && !(b->K.overlap_probability >= 1.0)) { td[0] = MYO_PI / 4.0; td[1] = MYO_PI / 4.0 - MYO_PI; }
      td[5] = m->site_pos[3 * b->K.target1_sid]; td[6] = m->site_pos[3 * b->K.target1_sid + 1];
      td[7] = m->site_pos[3 * b->K.target2_sid]; td[8] = m->site_pos[3 * b->K.target2_sid + 1];
      bd[0] = m->body_mass[b->K.obj1_bid]; bd[1] = m->body_mass[b->K.obj2_bid];
      for (int k = 0; k < 3; ++k) { bd[2 + k] = m->geom_friction[3 * b->K.obj1_gid + k]; bd[5 + k] = m->geom_friction[3 * b->K.obj2_gid + k]; }
      bd[8] = m->geom_size[3 * b->K.obj1_gid]; bd[9] = m->geom_size[3 * b->K.obj2_gid];
      if (b->K.task_choice == MYO_CHOICE_CW) mi[0] = MYO_WHICH_CW;
    }
  }
  void* p = nullptr;
  rc |= be_malloc(&p, host.size() * sizeof(double));
  if (!rc) rc |= be_h2d(p, host.data(), host.size() * sizeof(double));
  b->rec = (double*)p;
  b->allocs.push_back(p);
  {                                // health counters (myo_batch_health), zeroed
    void* hc = nullptr;
    const int zero4[4] = {0, 0, 0, 0};
    rc |= be_malloc(&hc, sizeof zero4);
    if (!rc) rc |= be_h2d(hc, zero4, sizeof zero4);
    b->K.health = (int*)hc;
    if (hc) b->allocs.push_back(hc);
  }
  if (dtype == MYO_F64) {          // the fp64 stepper keeps controls, moment arms and the warm start in global memory (ScratchPoses<double>::ctrl_g)
    rc |= be_batch_workspaces(b, n_envs, device);
  }
  b->K.objf_off = b->L.off_objfric - b->L.off_warm;      // (Scratch::SPILL reads the object group's friction in the record)
  if (m->integrator == 1) {        // RK4 stage storage, one RkScratch per env (global memory)
    void* w = nullptr;
    const size_t each = dtype == MYO_F64 ? sizeof(RkScratch<double>) : sizeof(RkScratch<float>);
    rc |= be_malloc(&w, each * (size_t)n_envs * MYO_PARTS_MAX);      // indexed by workgroup: a step in P parts launches P n of them
    b->K.rk_ws = w;
    if (w) b->allocs.push_back(w);
  }
  b->order = nullptr; b->cost = nullptr; b->ticks = nullptr; b->part_state = nullptr; b->step_gen = nullptr; b->plan.nparts = 1; b->plan.k[0] = 0; b->plan.k[1] = cfg ? cfg->frame_skip : 0; b->plan.wt = 0;
  {
    // the parts of an env step (k_step): MYO_STEP_SPLIT = "7,3" (substeps per part; "0" or one number = whole steps; A/B switch
    // of the tools and of the bit-identity tests).  The emulation build runs the parts one after the other through the record.
    StepPlan pl; pl.nparts = 0; pl.k[0] = 0;
    { const char* pm = getenv("MYO_PUBLISH"); pl.wt = pm && !strcmp(pm, "fence") ? 0 : 1; }     // A/B switch: "wt" (default) | "fence"
    if (b->K.kind && cfg->frame_skip >= 2) {
      const char* sp = getenv("MYO_STEP_SPLIT");
      if (sp) {
        int acc = 0;
        for (const char* q = sp; *q && pl.nparts < MYO_PARTS_MAX;) {
          const int v = atoi(q);
          if (v <= 0) break;
          acc += v; pl.k[++pl.nparts] = acc;
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
        if (acc != cfg->frame_skip) pl.nparts = 0;
      } else {
        // parts of decreasing length, 4 : 3 : 2 : 1 of the frame_skip (measured at 4096 envs, frame_skip 10, k_step ms:
        // whole 2.73, 7+3 2.33, 6+3+1 2.27, 5+3+2 2.27, 4+3+2+1 2.24, 5+3+1+1 2.25, 3+3+2+1+1 2.26)
        const int f = cfg->frame_skip, np = f < 4 ? f : 4;
        static const int w[4] = {4, 3, 2, 1};
        int wsum = 0, acc = 0;
        for (int i = 0; i < np; ++i) wsum += w[i];
        for (int i = 0; i < np; ++i) {
          int len = i == np - 1 ? f - acc : (f * w[i] + wsum / 2) / wsum;
          const int room = f - acc - (np - 1 - i);        // every later part keeps at least one substep
          len = len < 1 ? 1 : (len > room ? room : len);
          acc += len; pl.k[i + 1] = acc;
        }
        pl.nparts = np;
      }
    }
    if (pl.nparts >= 2) b->plan = pl;
  }
  rc = be_batch_launch_state(b, m, n_envs, rc);      // (the HIP backend: timing events, launch order, the step plan's state, the wrap census)
  if (rc) {
    int r2 = fail(MYO_E_DEVICE, "device allocation/upload failed: %s", be_errstr(rc));
    for (void* q : b->allocs) be_free(q);
    be_batch_release(b, device, 0);
    delete b;
    return r2;
  }
  *out = b;
  return MYO_OK;
}

extern "C" void myo_batch_destroy(myo_batch* b) {
  if (!b) return;
  be_batch_release(b, b->device, 1);
  for (void* q : b->allocs) be_free(q);
  delete b;
}
extern "C" int myo_batch_num_envs(const myo_batch* b) { return b ? b->n : -1; }
extern "C" int myo_batch_obs_dim(const myo_batch* b) { return b ? b->nobs : -1; }
extern "C" int myo_batch_lds_bytes(const myo_batch* b) {
  if (!b) return -1;
  const bool big = b->ncap > MYO_NCON_MAX;
  if (b->dtype == MYO_F64) return big ? (int)sizeof(Scratch<double, MYO_NCON_BIG>) : (int)sizeof(Scratch<double, MYO_NCON_F64>);
  if (big) return (int)sizeof(Scratch<float, MYO_NCON_BIG>);
  // mixed stepper, base capacity: an RK4 model keeps its stage storage behind the scratch (MYO_RK_IN_LDS); elsewhere it is in global memory
  return (int)(((sizeof(Scratch<float>) + 15) / 16 * 16) * (b->integrator == 1 ? 1 : 0) + (b->integrator == 1 ? sizeof(RkScratch<float>) : sizeof(Scratch<float>)));
}
extern "C" int myo_batch_dump_size(const myo_batch* b) { return b ? b->D.total : -1; }
extern "C" int myo_batch_dump_offset(const myo_batch* b, const char* n) {
  if (!b || !n) return -1;
#define S(x) if (!strcmp(n, #x)) return b->D.x;
  S(ten_length) S(ten_J) S(M) S(qfrc_bias) S(qfrc_passive) S(qfrc_actuator) S(qacc_smooth) S(qacc) S(actuator_force)
  S(act_dot) S(counts) S(efc_aref) S(efc_D) S(site_xpos) S(subtree_com) S(xpos)
#undef S
  return -1;
}

