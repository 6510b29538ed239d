// myo_mjb.h — host-side reader of MuJoCo 2.1 binary models (.mjb) for libmyobatch: myo_model_load_mjb().
//
// Replaces what `gym.make(id, model_path=…/myo_hand_baoding.mjb)` does in the reference through MuJoCo's mj_loadModel
// (/root/reference/src/envs/__init__.py:17,63).  File layout: 4 header ints, 57 size ints, the
// mjOption block, 608 bytes of mjVisual + mjStatistic, then nbuffer bytes with every array of MJMODEL_POINTERS in order,
// each aligned to its element size (the table is csrc/mjb_layout.inc, shared with myochallenge_amd/mjb.py).  The decoded
// arrays plus the derived fields of myochallenge_amd/model.py:compile_model (tree depths, candidate collision pairs)
// are put into the model-blob format of include/myo_model_blob.h and handed to myo_model_from_blob, so both entry points
// build the same myo_model.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/myo_model_blob.h"

namespace myo_mjb {

static const char* const kLayout =
#include "mjb_layout.inc"
    ;

static const char* const kSizeNames[57] = {
    "nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ncam", "nlight", "nmesh", "nmeshvert", "nmeshtexvert", "nmeshface",
    "nmeshgraph", "nskin", "nskinvert", "nskintexvert", "nskinface", "nskinbone", "nskinbonevert", "nhfield", "nhfielddata", "ntex",
    "ntexdata", "nmat", "npair", "nexclude", "neq", "ntendon", "nwrap", "nsensor", "nnumeric", "nnumericdata", "ntext", "ntextdata",
    "ntuple", "ntupledata", "nkey", "nmocap", "nuser_body", "nuser_jnt", "nuser_geom", "nuser_site", "nuser_cam", "nuser_tendon",
    "nuser_actuator", "nuser_sensor", "nnames", "nM", "nemax", "njmax", "nconmax", "nstack", "nuserdata", "nsensordata", "nbuffer"};

struct File {
  std::map<std::string, long long> sizes;
  double timestep, impratio, tolerance, gravity[3], o_margin, meaninertia;
  double wind[3], density, viscosity;
  int integrator, collision, cone, iterations, disableflags, enableflags, solver, noslip_iterations;
  std::map<std::string, std::vector<double>> d;   // f64 arrays (row-major)
  std::map<std::string, std::vector<int>> i;      // i32 and u8 arrays
};

static long long dim(const std::string& expr, const std::map<std::string, long long>& sizes, bool* ok) {
  long long out = 1;
  size_t p = 0;
  while (p <= expr.size()) {
    size_t q = expr.find('*', p);
    const std::string tok = expr.substr(p, q == std::string::npos ? std::string::npos : q - p);
    if (!tok.empty() && tok.find_first_not_of("0123456789") == std::string::npos) out *= atoll(tok.c_str());
    else {
      auto it = sizes.find(tok);
      if (it == sizes.end()) { *ok = false; return 0; }
      out *= it->second;
    }
    if (q == std::string::npos) break;
    p = q + 1;
  }
  return out;
}

static bool parse(const unsigned char* blob, size_t n, File& f, std::string& err) {
  char msg[256];
  if (n < 16 + 4 * 57) { err = "file too short for an MJB header"; return false; }
  int head[4];
  memcpy(head, blob, 16);
  if (head[0] != 54321) { snprintf(msg, sizeof msg, "bad magic %d (want 54321)", head[0]); err = msg; return false; }
  if (head[1] != 8) { err = "sizeof(mjtNum) != 8: only double-precision models are supported"; return false; }
  if (head[2] != 57 || head[3] != 266) {
    snprintf(msg, sizeof msg, "header says %d sizes / %d pointers; this reader handles MuJoCo 2.1 (57 / 266)", head[2], head[3]);
    err = msg; return false;
  }
  size_t off = 16;
  for (int k = 0; k < 57; ++k) {
    int v; memcpy(&v, blob + off, 4); off += 4; f.sizes[kSizeNames[k]] = v;
    if (v < 0) { snprintf(msg, sizeof msg, "negative size %s = %d in the header", kSizeNames[k], v); err = msg; return false; }
  }
  // mjOption: timestep apirate impratio tolerance noslip_tolerance mpr_tolerance gravity[3] wind[3] magnetic[3] density viscosity
  //           o_margin o_solref[2] o_solimp[5] | integrator collision cone jacobian solver iterations noslip_it mpr_it disable enable
  double od[25];
  if (off + sizeof od + 40 > n) { err = "file too short for mjOption"; return false; }
  memcpy(od, blob + off, sizeof od); off += sizeof od;
  f.timestep = od[0]; f.impratio = od[2]; f.tolerance = od[3];
  f.gravity[0] = od[6]; f.gravity[1] = od[7]; f.gravity[2] = od[8]; f.o_margin = od[17];
  f.wind[0] = od[9]; f.wind[1] = od[10]; f.wind[2] = od[11]; f.density = od[15]; f.viscosity = od[16];
  int oi[10];
  memcpy(oi, blob + off, sizeof oi); off += sizeof oi;
  f.integrator = oi[0]; f.collision = oi[1]; f.cone = oi[2]; f.iterations = oi[5]; f.disableflags = oi[8]; f.enableflags = oi[9];
  f.solver = oi[4]; f.noslip_iterations = oi[6];
  const long long nbuffer = f.sizes["nbuffer"];
  if (nbuffer < 0 || (size_t)nbuffer > n || n - (size_t)nbuffer < off) { err = "nbuffer larger than the file"; return false; }
  const size_t base = n - (size_t)nbuffer;
  if (base - off != 608) {
    snprintf(msg, sizeof msg, "expected 608 bytes of mjVisual + mjStatistic before the array buffer (MuJoCo 2.1), found %zu", base - off);
    err = msg; return false;
  }
  memcpy(&f.meaninertia, blob + base - 56, 8);
  // arrays
  size_t pos = 0;
  const char* p = kLayout;
  while (*p) {
    const char* e = strchr(p, '\n');
    std::string line(p, e ? (size_t)(e - p) : strlen(p));
    p = e ? e + 1 : p + line.size();
    if (line.empty()) continue;
    char t; char name[64], rows[64], cols[64];
    if (sscanf(line.c_str(), " %c %63s %63s %63s", &t, name, rows, cols) != 4) { err = "bad layout line: " + line; return false; }
    bool ok = true;
    const long long nr = dim(rows, f.sizes, &ok), nc = dim(cols, f.sizes, &ok);
    if (!ok || nr < 0 || nc < 0) { err = std::string("unknown size in layout of ") + name; return false; }
    const size_t isz = (t == 'd') ? 8 : (t == 'i' || t == 'f') ? 4 : 1;
    // every count is bounded by the file size BEFORE anything is multiplied (sizes are attacker-controlled 31-bit numbers)
    if ((nr && (unsigned long long)nc > (unsigned long long)n / (unsigned long long)nr) || (unsigned long long)(nr * nc) > (unsigned long long)n / isz) {
      err = std::string("array ") + name + " is larger than the file"; return false;
    }
    const long long cnt = nr * nc;
    if (cnt) pos = (pos + isz - 1) / isz * isz;
    if (base + pos > n || (size_t)cnt * isz > n - (base + pos)) { err = std::string("array ") + name + " runs past the end of the file"; return false; }
    const unsigned char* src = blob + base + pos;
    if (t == 'd') { std::vector<double>& v = f.d[name]; v.resize((size_t)cnt); if (cnt) memcpy(v.data(), src, (size_t)cnt * 8); }
    else if (t == 'i') { std::vector<int>& v = f.i[name]; v.resize((size_t)cnt); if (cnt) memcpy(v.data(), src, (size_t)cnt * 4); }
    else if (t == 'b') { std::vector<int>& v = f.i[name]; v.resize((size_t)cnt); for (long long k = 0; k < cnt; ++k) v[(size_t)k] = src[k]; }
    pos += (size_t)cnt * isz;      // f32 / char arrays (meshes, textures, names) are skipped: nothing in the stepper reads them
  }
  if ((long long)pos != nbuffer) {
    snprintf(msg, sizeof msg, "decoded %zu bytes of arrays but nbuffer = %lld: unsupported layout", pos, nbuffer);
    err = msg; return false;
  }
  return true;
}

// the fields of a compiled model (myochallenge_amd/model.py: _INT_FIELDS, _F64_FIELDS)
static const char* const kIntFields[] = {
    "body_parentid", "body_rootid", "body_weldid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "jnt_type", "jnt_qposadr",
    "jnt_dofadr", "jnt_bodyid", "jnt_limited", "dof_bodyid", "dof_jntid", "dof_parentid", "geom_type", "geom_contype", "geom_conaffinity",
    "geom_condim", "geom_bodyid", "geom_priority", "site_bodyid", "tendon_adr", "tendon_num", "tendon_limited", "wrap_type", "wrap_objid",
    "actuator_trntype", "actuator_dyntype", "actuator_gaintype", "actuator_biastype", "actuator_trnid", "actuator_ctrllimited",
    "actuator_forcelimited"};
static const char* const kF64Fields[] = {
    "qpos0", "qpos_spring", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_invweight0", "jnt_solref",
    "jnt_solimp", "jnt_pos", "jnt_axis", "jnt_stiffness", "jnt_range", "jnt_margin", "dof_armature", "dof_damping", "dof_invweight0",
    "geom_solmix", "geom_solref", "geom_solimp", "geom_size", "geom_rbound", "geom_pos", "geom_quat", "geom_friction", "geom_margin",
    "geom_gap", "site_pos", "site_quat", "tendon_solref_lim", "tendon_solimp_lim", "tendon_range", "tendon_margin", "tendon_stiffness", "tendon_damping",
    "tendon_lengthspring", "tendon_invweight0", "wrap_prm", "actuator_dynprm", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange",
    "actuator_forcerange", "actuator_gear", "actuator_acc0", "actuator_lengthrange",
    "dof_frictionloss", "dof_solref", "dof_solimp", "tendon_frictionloss", "tendon_solref_fri", "tendon_solimp_fri"};

static bool pair_supported(int t1, int t2) {       // narrow phases of csrc/myo_physics.h:collide_pair (types ordered t1 <= t2)
  return (t1 == MYO_GEOM_PLANE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE || t2 == MYO_GEOM_ELLIPSOID || t2 == MYO_GEOM_CYLINDER)) ||
         (t1 == MYO_GEOM_SPHERE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE || t2 == MYO_GEOM_BOX || t2 == MYO_GEOM_CYLINDER || t2 == MYO_GEOM_ELLIPSOID)) ||
         (t1 == MYO_GEOM_CAPSULE && (t2 == MYO_GEOM_CAPSULE || t2 == MYO_GEOM_BOX || t2 == MYO_GEOM_CYLINDER || t2 == MYO_GEOM_ELLIPSOID)) ||
         (t1 == MYO_GEOM_BOX && t2 == MYO_GEOM_BOX);
}

// compile_model(): feature checks, derived fields, blob.  integrator < 0 keeps the model's; allow_drop = 0 refuses a model that
// has colliding geom pairs without a narrow phase here (MYO_E_UNSUPPORTED at the caller), 1 compiles without them.
static bool to_blob(File& f, int integrator, int allow_flags, std::vector<unsigned char>& out, std::string& err, int* unsupported) {
  // allow_flags (myo_model_load_mjb's `unsupported_contacts`): bit 0 = compile without the colliding pairs that have no narrow phase;
  // bit 1 = step a model whose opt.solver is PGS / CG, or that asks for noslip iterations, with THIS solver (Newton, no noslip pass) — the
  // caller's explicit choice (the reference's .mjb files are not available here, and a refusal without an override would
  // leave their user no way on); everything else on the list stays refused
  const int allow_drop = allow_flags & 1, allow_solver = (allow_flags >> 1) & 1;
  auto I = [&](const char* k) -> std::vector<int>& { return f.i[k]; };
  auto D = [&](const char* k) -> std::vector<double>& { return f.d[k]; };
  const int nbody = (int)f.sizes["nbody"], njnt = (int)f.sizes["njnt"], ngeom = (int)f.sizes["ngeom"], nv = (int)f.sizes["nv"];
  const int ntendon = (int)f.sizes["ntendon"], nsite = (int)f.sizes["nsite"], nwrap = (int)f.sizes["nwrap"];
  // ---- the file is untrusted: every id array this function indexes with is range-checked first (myo_model_from_blob
  // checks the rest); a corrupt model is MYO_E_ARG, never an out-of-bounds read
  {
    char m[160];
    auto bad = [&](const char* what, int i, int v) { snprintf(m, sizeof m, "corrupt model: %s[%d] = %d is out of range", what, i, v); err = m; return false; };
    if (nbody < 1) { err = "corrupt model: nbody < 1"; return false; }
    for (int b = 0; b < nbody; ++b) {
      const int pa = I("body_parentid")[b], w = I("body_weldid")[b];
      if (pa < 0 || (b > 0 && pa >= b) || (b == 0 && pa != 0)) return bad("body_parentid", b, pa);
      if (w < 0 || w >= nbody) return bad("body_weldid", b, w);
    }
    for (int g = 0; g < ngeom; ++g) {
      if (I("geom_bodyid")[g] < 0 || I("geom_bodyid")[g] >= nbody) return bad("geom_bodyid", g, I("geom_bodyid")[g]);
      if (I("geom_type")[g] < 0 || I("geom_type")[g] > MYO_GEOM_MESH) return bad("geom_type", g, I("geom_type")[g]);
    }
    for (int k = 0; k < nsite; ++k) if (I("site_bodyid")[k] < 0 || I("site_bodyid")[k] >= nbody) return bad("site_bodyid", k, I("site_bodyid")[k]);
    for (int i = 0; i < nv; ++i) if (I("dof_parentid")[i] < -1 || I("dof_parentid")[i] >= i) return bad("dof_parentid", i, I("dof_parentid")[i]);
    for (int t = 0; t < ntendon; ++t) {
      const int a = I("tendon_adr")[t], c = I("tendon_num")[t];
      if (a < 0 || c < 0 || a > nwrap || c > nwrap - a) return bad("tendon_adr/num", t, a);
    }
    for (int w = 0; w < nwrap; ++w) {
      const int wt = I("wrap_type")[w], id = I("wrap_objid")[w];
      if (wt == MYO_WRAP_SITE && (id < 0 || id >= nsite)) return bad("wrap_objid (site)", w, id);
      if ((wt == MYO_WRAP_SPHERE || wt == MYO_WRAP_CYLINDER)) {
        if (id < 0 || id >= ngeom) return bad("wrap_objid (geom)", w, id);
        const double sp = D("wrap_prm")[w];
        if (sp >= 0 && !(sp < (double)nsite)) return bad("wrap_prm (side site)", w, (int)sp);
      }
    }
  }
  // what mj_collision / mj_step would do differently and this stepper does not restate is refused, never ignored — and ALL of it is
  // reported at once (model.py:unsupported_features builds the same list; tests/test_mjb_and_model.py compares the two routes)
  std::vector<std::string> uns;
  auto U = [&](const char* fmt, long long a = 0, long long b = 0) { char m[256]; snprintf(m, sizeof m, fmt, a, b); uns.push_back(m); };
  const int npair_x = f.collision != 2 ? (int)f.sizes["npair"] : 0;      // explicit <contact><pair> entries in force (opt.collision: 0 all, 1 predefined, 2 dynamic)
  {
    int n_aniso = 0, n_dim = 0;
    for (int k = 0; k < npair_x; ++k) {
      const double* fr = &D("pair_friction")[5 * (size_t)k];
      const int dim = I("pair_dim")[k];
      if (fr[0] != fr[1] || fr[3] != fr[4]) n_aniso++;
      if (dim != 1 && dim != 3 && dim != 4 && dim != 6) n_dim++;
      if (I("pair_geom1")[k] < 0 || I("pair_geom1")[k] >= ngeom || I("pair_geom2")[k] < 0 || I("pair_geom2")[k] >= ngeom) { err = "an explicit contact pair names a geom out of range"; return false; }
    }
    if (n_aniso) U("[pair_anisotropic x%lld] explicit contact pair(s) with anisotropic friction are not supported", n_aniso);
    if (n_dim) U("[pair_condim x%lld] explicit contact pair(s) with a condim other than 1, 3, 4, 6", n_dim);
  }
  if (f.disableflags & ~((1 << 9) | (1 << 11))) U("[disableflags x1] opt.disableflags = %#llx: only filterparent and refsafe can be disabled in this stepper", f.disableflags);
  if (f.enableflags & 1) U("[override x1] opt.enableflags: contact override is not supported");
  { int n = 0; for (int j = 0; j < njnt; ++j) n += I("jnt_type")[j] == MYO_JNT_BALL; if (n) U("[ball_joints x%lld] ball joints are not supported", n); }
  if (f.sizes["neq"] > 0) U("[equality x%lld] equality constraints are not supported", f.sizes["neq"]);
  if (f.cone != 0) U("[cone x1] only pyramidal friction cones are supported (opt.cone = elliptic)");
  if (f.solver != 2 && !allow_solver) U("[solver x1] opt.solver = %lld: this stepper restates mj_solNewton only (a model asking for PGS / CG would be stepped with another algorithm)", f.solver);
  if (f.noslip_iterations > 0 && !allow_solver) U("[noslip x1] opt.noslip_iterations = %lld: the noslip post-solver is not implemented", f.noslip_iterations);
  {
    const int nf = (f.density != 0.0) + (f.viscosity != 0.0) + (f.wind[0] != 0.0 || f.wind[1] != 0.0 || f.wind[2] != 0.0);
    if (nf) U("[fluid x%lld] opt.density / viscosity / wind non-zero: fluid forces in mj_passive are not implemented", nf);
  }
  if (f.integrator != 0 && f.integrator != 1) U("[integrator x1] opt.integrator = %lld: Euler (0) and RK4 (1) are implemented", f.integrator);
  {
    int n = 0;
    for (int g = 0; g < ngeom; ++g)
      if ((I("geom_contype")[g] | I("geom_conaffinity")[g]) != 0) { const int cd = I("geom_condim")[g]; n += (cd != 1 && cd != 3 && cd != 4 && cd != 6); }
    if (n) U("[condim x%lld] contact dimensions (condim) other than 1, 3, 4, 6 do not exist in MuJoCo", n);
  }
  {
    const int nu = (int)f.sizes["nu"];
    int n_trn = 0, n_dyn = 0, n_gain = 0, n_bias = 0;
    for (int i = 0; i < nu; ++i) {
      n_trn += I("actuator_trntype")[i] != 3;
      const int dy = I("actuator_dyntype")[i], ga = I("actuator_gaintype")[i], bi = I("actuator_biastype")[i];
      n_dyn += !(dy == 0 || dy == 3); n_gain += !(ga == 0 || ga == 1); n_bias += !(bi >= 0 && bi <= 2);
    }
    if (n_trn) U("[transmission x%lld] only tendon transmissions are supported", n_trn);
    if (n_dyn) U("[actuator_dyn x%lld] actuator dyntype integrator / filter / user is not implemented (none and muscle are)", n_dyn);
    if (n_gain) U("[actuator_gain x%lld] actuator gaintype user is not implemented (fixed and muscle are)", n_gain);
    if (n_bias) U("[actuator_bias x%lld] actuator biastype user is not implemented (none, affine and muscle are)", n_bias);
  }
  // static collision filter (mj_collision body-pair pass): same weld group, <exclude> body pairs (exclude_signature =
  // ((body1 + 1) << 16) + body2 + 1 with body1 < body2, MuJoCo 2.1), parent-child unless mjDSBL_FILTERPARENT, contype / conaffinity
  const std::vector<int>& excl = I("exclude_signature");
  const bool filterparent = !(f.disableflags & (1 << 9));
  std::vector<int> p1, p2, psub, pxp;
  int dropped = 0;
  char msg[256] = "";
  const std::vector<int>&gb = I("geom_bodyid"), &weld = I("body_weldid"), &par = I("body_parentid"), &gt = I("geom_type");
  auto emit = [&](int ga, int gb_, int xp) {
    const int t1 = gt[ga], t2 = gt[gb_];
    const int a = t1 <= t2 ? ga : gb_, b = t1 <= t2 ? gb_ : ga;      // MuJoCo orders a pair by geom type
    const int lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
    if (lo == MYO_GEOM_PLANE && hi == MYO_GEOM_PLANE) return;
    if (lo == MYO_GEOM_BOX && hi == MYO_GEOM_BOX) { for (int v = 0; v < 17; ++v) { p1.push_back(a); p2.push_back(b); psub.push_back(1 + v); pxp.push_back(xp); } }   // 16 vertex-face candidates + the edge-edge candidate (17)
    else if (pair_supported(lo, hi)) { p1.push_back(a); p2.push_back(b); psub.push_back(0); pxp.push_back(xp); }
    else { if (!dropped) snprintf(msg, sizeof msg, "geom %d (type %d) - geom %d (type %d)", a, gt[a], b, gt[b]); dropped++; }
  };
  // explicit pairs first (no contype / conaffinity / parent / exclude filtering applies to them).  mj_collision merges them into its
  // BODY-pair sweep by pair_signature (the key of exclude_signature): a body pair that has explicit pairs gets only those — every dynamic
  // geom pair between the two bodies is skipped (model.py:collision_pairs)
  std::vector<long long> xsig;
  for (int k = 0; k < npair_x; ++k) {
    const int a = I("pair_geom1")[k], b = I("pair_geom2")[k];
    const int ba = gb[a] < gb[b] ? gb[a] : gb[b], bb = gb[a] < gb[b] ? gb[b] : gb[a];
    xsig.push_back(((long long)ba << 32) | (unsigned)bb);
    emit(a, b, k);
  }
  for (int g1 = 0; g1 < ngeom && f.collision != 1; ++g1)
    for (int g2 = g1 + 1; g2 < ngeom; ++g2) {
      if (!xsig.empty()) {
        const int ba = gb[g1] < gb[g2] ? gb[g1] : gb[g2], bb = gb[g1] < gb[g2] ? gb[g2] : gb[g1];
        if (std::find(xsig.begin(), xsig.end(), (((long long)ba << 32) | (unsigned)bb)) != xsig.end()) continue;
      }
      const int w1 = weld[gb[g1]], w2 = weld[gb[g2]];
      if (w1 == w2) continue;
      if (!excl.empty()) {
        const int ba = gb[g1] < gb[g2] ? gb[g1] : gb[g2], bb = gb[g1] < gb[g2] ? gb[g2] : gb[g1];
        const int sig = ((ba + 1) << 16) + bb + 1;
        bool hit = false;
        for (int x : excl) if (x == sig) hit = true;
        if (hit) continue;
      }
      const int wp1 = weld[par[w1]], wp2 = weld[par[w2]];
      if (filterparent && w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
      if (!((I("geom_contype")[g1] & I("geom_conaffinity")[g2]) || (I("geom_contype")[g2] & I("geom_conaffinity")[g1]))) continue;
      emit(g1, g2, -1);
    }
  {   // stable partition: the pairs of the primitive narrow phases first (model.py:compile_model does the same)
    auto is_std = [&](size_t k) {
      const int t1 = gt[p1[k]], t2 = gt[p2[k]];
      return (t1 == MYO_GEOM_PLANE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE)) || (t1 == MYO_GEOM_SPHERE && (t2 == MYO_GEOM_SPHERE || t2 == MYO_GEOM_CAPSULE || t2 == MYO_GEOM_BOX)) ||
             (t1 == MYO_GEOM_CAPSULE && t2 == MYO_GEOM_CAPSULE);
    };
    std::vector<int> q1, q2, qs, qx;
    for (int pass = 0; pass < 2; ++pass)
      for (size_t k = 0; k < p1.size(); ++k)
        if (is_std(k) == (pass == 0)) { q1.push_back(p1[k]); q2.push_back(p2[k]); qs.push_back(psub[k]); qx.push_back(pxp[k]); }
    p1.swap(q1); p2.swap(q2); psub.swap(qs); pxp.swap(qx);
  }
  if (dropped && !allow_drop) {
    char m2[400];
    snprintf(m2, sizeof m2, "[contact_pairs x%d] %d colliding geom pair(s) have no narrow phase in this stepper (first: %s); load with unsupported_contacts = 1 to compile without them", dropped, dropped, msg);
    uns.push_back(m2);
  }
  int n_fixed = 0, n_inside = 0;
  for (int t = 0; t < ntendon; ++t)
    for (int w = I("tendon_adr")[t]; w < I("tendon_adr")[t] + I("tendon_num")[t]; ++w) {
      const int wt = I("wrap_type")[w];
      if (wt == MYO_WRAP_JOINT) { n_fixed++; continue; }
      if ((wt == MYO_WRAP_SPHERE || wt == MYO_WRAP_CYLINDER) && D("wrap_prm")[w] >= 0) {
        const int sid = (int)lround(D("wrap_prm")[w]), gid = I("wrap_objid")[w];
        if (I("site_bodyid")[sid] == gb[gid]) {      // a side site inside its wrap geom needs MuJoCo's inside-wrap iteration
          double d[3];
          for (int k = 0; k < 3; ++k) d[k] = D("site_pos")[3 * sid + k] - D("geom_pos")[3 * gid + k];
          if (wt == MYO_WRAP_CYLINDER) {
            const double* q = &D("geom_quat")[4 * gid];
            const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const double w_ = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
            const double ax[3] = {2 * (x * z + w_ * y), 2 * (y * z - w_ * x), w_ * w_ - x * x - y * y + z * z};   // cylinder axis (3rd column of R)
            const double along = d[0] * ax[0] + d[1] * ax[1] + d[2] * ax[2];
            for (int k = 0; k < 3; ++k) d[k] -= along * ax[k];
          }
          if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < D("geom_size")[3 * gid]) n_inside++;
        }
      }
    }
  if (n_fixed) U("[fixed_tendons x%lld] fixed (joint) tendons are not supported", n_fixed);
  if (n_inside) U("[side_site_inside x%lld] a wrapping side site lies inside its wrap geom (MuJoCo's inside-wrap iteration is not implemented)", n_inside);
  {
    int n = 0;
    const std::vector<double>& fl = D("dof_frictionloss");
    for (int d = 0; d < nv && (size_t)d < fl.size(); ++d) { const int j = I("dof_jntid")[d]; n += fl[d] > 0 && j >= 0 && j < njnt && I("jnt_type")[j] == MYO_JNT_FREE; }
    if (n) U("[frictionloss_free x%lld] friction loss on the dofs of a free joint is not supported", n);
  }
  if (!uns.empty()) {
    err.clear();
    for (size_t k = 0; k < uns.size(); ++k) { if (k) err += "; "; err += uns[k]; }
    *unsupported = 1;
    return false;
  }
  // fields
  std::vector<std::pair<std::string, std::vector<int>>> fi;
  std::vector<std::pair<std::string, std::vector<double>>> fd;
  fi.push_back({"sizes", {(int)f.sizes["nq"], nv, (int)f.sizes["nu"], (int)f.sizes["na"], nbody, njnt, ngeom, (int)f.sizes["nsite"], ntendon,
                          (int)f.sizes["nwrap"]}});
  for (const char* k : kIntFields) fi.push_back({k, I(k)});
  for (const char* k : kF64Fields) fd.push_back({k, D(k)});
  std::vector<int> bd(nbody, 0), dd(nv, 0);
  for (int b = 1; b < nbody; ++b) bd[b] = bd[par[b]] + 1;
  for (int i = 0; i < nv; ++i) { const int p = I("dof_parentid")[i]; dd[i] = p < 0 ? 1 : dd[p] + 1; }
  fi.push_back({"x_body_depth", bd}); fi.push_back({"x_dof_depth", dd});
  fi.push_back({"x_pair_geom1", p1}); fi.push_back({"x_pair_geom2", p2}); fi.push_back({"x_pair_sub", psub});
  {   // explicit <pair> parameters (model.py:compile_model writes the same fields)
    std::vector<int> xdim;
    std::vector<double> xmargin, xgap, xsolref, xsolimp, xfric;
    for (int k = 0; k < npair_x; ++k) {
      xdim.push_back(I("pair_dim")[k]); xmargin.push_back(D("pair_margin")[k]); xgap.push_back(D("pair_gap")[k]);
      for (int e = 0; e < 2; ++e) xsolref.push_back(D("pair_solref")[2 * (size_t)k + e]);
      for (int e = 0; e < 5; ++e) xsolimp.push_back(D("pair_solimp")[5 * (size_t)k + e]);
      const double* fr = &D("pair_friction")[5 * (size_t)k];
      xfric.push_back(fr[0]); xfric.push_back(fr[2]); xfric.push_back(fr[3]);
    }
    fi.push_back({"x_pair_explicit", pxp}); fi.push_back({"x_xp_dim", xdim});
    fd.push_back({"x_xp_margin", xmargin}); fd.push_back({"x_xp_gap", xgap}); fd.push_back({"x_xp_solref", xsolref});
    fd.push_back({"x_xp_solimp", xsolimp}); fd.push_back({"x_xp_friction", xfric});
  }
  fi.push_back({"opt_int", {integrator >= 0 ? integrator : f.integrator, f.cone, f.iterations, f.disableflags}});
  fd.push_back({"opt_f64", {f.timestep, f.tolerance, f.impratio, f.gravity[0], f.gravity[1], f.gravity[2], f.o_margin, f.meaninertia}});
  const size_t nf = fi.size() + fd.size();
  size_t head = sizeof(myo_blob_header) + nf * sizeof(myo_blob_field);
  head = (head + 7) / 8 * 8;
  std::vector<myo_blob_field> table;
  std::vector<unsigned char> payload;
  auto add = [&](const std::string& name, uint32_t dt, const void* data, size_t count, size_t isz) {
    myo_blob_field e;
    memset(&e, 0, sizeof e);
    strncpy(e.name, name.c_str(), MYO_BLOB_NAME_LEN - 1);
    e.dtype = dt; e.count = (uint32_t)count; e.offset = head + payload.size();
    table.push_back(e);
    const size_t nb = count * isz;
    payload.insert(payload.end(), (const unsigned char*)data, (const unsigned char*)data + nb);
    payload.resize((payload.size() + 7) / 8 * 8, 0);
  };
  for (auto& kv : fi) add(kv.first, MYO_BLOB_I32, kv.second.data(), kv.second.size(), 4);
  for (auto& kv : fd) add(kv.first, MYO_BLOB_F64, kv.second.data(), kv.second.size(), 8);
  myo_blob_header h;
  h.magic = MYO_BLOB_MAGIC; h.version = MYO_BLOB_VERSION; h.n_fields = (uint32_t)nf; h.total_bytes = (uint32_t)(head + payload.size());
  out.assign(head + payload.size(), 0);
  memcpy(out.data(), &h, sizeof h);
  memcpy(out.data() + sizeof h, table.data(), table.size() * sizeof(myo_blob_field));
  memcpy(out.data() + head, payload.data(), payload.size());
  return true;
}

}  // namespace myo_mjb
