/* myo_oracle.c — TEST INFRASTRUCTURE (see myo_oracle.h for the parity status).
 *
 * Scalar fp64 restatement of MuJoCo 2.1's mj_step pipeline for the feature subset the
 * MyoSuite muscle-tendon models use, plus MyoSuite's Baoding task step.  Written from the
 * published algorithm; the code the reference actually runs lives in
 * un-vendored third-party packages (MuJoCo 2.1.x via free-mujoco-py==2.1.6, MyoSuite==1.2.3;
 * /root/reference/requirements.txt:41,81).  Stage names follow SURVEY.md §8a P1-P12.
 */
#include "myo_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/myo_model_blob.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MAXCON 64
#define MAXEFC 320
#define LS_ITER 50
#define LS_TOL 0.01

struct OrcModel {
  void* blob;
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, npair;
  int integrator, cone, iterations, disableflags;
  double timestep, tolerance, impratio, gravity[3], o_margin, meaninertia;
  const int *body_parentid, *body_rootid, *body_weldid, *body_jntnum, *body_jntadr, *body_dofnum,
      *body_dofadr, *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited, *dof_bodyid,
      *dof_jntid, *dof_parentid, *geom_type, *geom_condim, *geom_bodyid, *geom_priority,
      *site_bodyid, *tendon_adr, *tendon_num, *tendon_limited, *wrap_type, *wrap_objid,
      *actuator_trntype, *actuator_dyntype, *actuator_gaintype, *actuator_biastype,
      *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited, *pair_geom1, *pair_geom2, *pair_sub;
  /* explicit <contact><pair> entries (optional blob fields): per pair row the entry's index (-1: a dynamic pair); the entries */
  const int *pair_explicit, *xp_dim;
  const double *xp_margin, *xp_gap, *xp_solref, *xp_solimp, *xp_friction;
  const double *qpos0, *qpos_spring, *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass,
      *body_inertia, *body_invweight0, *jnt_solref, *jnt_solimp, *jnt_pos, *jnt_axis,
      *jnt_stiffness, *jnt_range, *jnt_margin, *dof_armature, *dof_damping, *dof_invweight0,
      *geom_solmix, *geom_solref, *geom_solimp, *geom_size, *geom_rbound, *geom_pos, *geom_quat,
      *geom_friction, *geom_margin, *geom_gap, *site_pos, *tendon_solref_lim, *tendon_solimp_lim,
      *tendon_range, *tendon_margin, *tendon_stiffness, *tendon_damping, *tendon_lengthspring,
      *tendon_invweight0, *wrap_prm, *actuator_dynprm, *actuator_gainprm, *actuator_biasprm,
      *actuator_ctrlrange, *actuator_forcerange, *actuator_gear, *actuator_acc0,
      *actuator_lengthrange;
  /* friction loss (optional blob fields; NULL = none): dof_frictionloss nv, dof_solref 2 nv, dof_solimp 5 nv,
   * tendon_frictionloss ntendon, tendon_solref_fri 2 ntendon, tendon_solimp_fri 5 ntendon */
  const double *dof_frictionloss, *dof_solref, *dof_solimp, *tendon_frictionloss, *tendon_solref_fri, *tendon_solimp_fri;
};

typedef struct {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5];
  int geom1, geom2, dim;      /* dim: condim of the contact (1, 3, 4 or 6) */
} OrcContact;

typedef struct { const char* name; double* p; int n; } OrcField;

struct OrcData {
  const OrcModel* m;
  double time;
  int ncon, nefc, solver_iter, bad, nl, ntl;
  /* state */
  double *qpos, *qvel, *act, *ctrl, *qacc_warmstart;
  /* per-env model overrides (P2 randomisation, reference baoding.py:559-604) */
  double *body_mass, *geom_friction, *geom_size, *geom_pos, *site_pos;
  /* position stage */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat,
      *site_xpos, *subtree_com, *cinert, *cdof, *crb, *ten_length, *ten_J, *actuator_length,
      *actuator_moment, *M, *Mchol;
  /* velocity stage */
  double *ten_velocity, *actuator_velocity, *cvel, *cdof_dot, *qfrc_passive, *qfrc_bias;
  /* actuation / acceleration */
  double *act_dot, *actuator_force, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qacc,
      *qfrc_constraint;
  /* constraints */
  OrcContact con[MAXCON];
  double *efc_J, *efc_pos, *efc_margin, *efc_D, *efc_R, *efc_aref, *efc_vel, *efc_force,
      *efc_diagApprox, *efc_KBIP, *efc_floss;      /* efc_floss: friction loss of a friction row (type 3), else 0 */
  int efc_type[MAXEFC], efc_id[MAXEFC];
  /* scratch */
  double *w1, *w2, *w3, *w4, *w5, *H, *cacc, *cfrc;
  OrcField fields[96];
  int nfields;
};

/* ------------------------------------------------------------------ flop counter ("counted, not guessed")
 * Every floating-point add, subtract, multiply, divide, compare-free min/max and sqrt of the algorithm AS WRITTEN
 * HERE (dense Jacobian and dense JᵀDJ, as MuJoCo's dense path for nv < 60) counts 1; a transcendental (sin, cos,
 * exp, atan2, pow) counts 1 as well.  Counted per stage; not thread-safe (one process = one counter). */
enum { ST_KIN, ST_COM, ST_TENDON, ST_CRB, ST_COLLIDE, ST_CONSTRAINT, ST_VELOCITY, ST_ACTUATION, ST_ACCEL, ST_NEWTON, ST_INTEGRATE, ST_TASK, ST_N };
static double g_flops[ST_N];
static int g_stage __attribute__((unused)) = ST_TASK;
#ifdef ORC_COUNT_FLOPS      /* libmyo_oracle_flops.so; the timed CPU baseline uses the build without the counter */
#define FL(n) (g_flops[g_stage] += (double)(n))
#define STAGE(x) (g_stage = (x))
#else
#define FL(n) ((void)0)
#define STAGE(x) ((void)0)
#endif
int orc_flops_enabled(void) {
#ifdef ORC_COUNT_FLOPS
  return 1;
#else
  return 0;
#endif
}
void orc_flops(double* out, int reset) {
  if (out) memcpy(out, g_flops, sizeof g_flops);
  if (reset) memset(g_flops, 0, sizeof g_flops);
}
int orc_flops_stages(void) { return ST_N; }

/* ------------------------------------------------------------------ small vector math */
static double dot3(const double* a, const double* b) { FL(5); return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  FL(9);
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  r[0]=x; r[1]=y; r[2]=z;
}
static double norm3(const double* a) { FL(1); return sqrt(dot3(a,a)); }
static double normalize3(double* a) {
  double n = norm3(a); FL(3);
  if (n < MINVAL) { a[0]=1; a[1]=0; a[2]=0; } else { a[0]/=n; a[1]/=n; a[2]/=n; }
  return n;
}
static void mulmatvec3(double* r, const double* R, const double* v) {
  FL(15);
  double x=R[0]*v[0]+R[1]*v[1]+R[2]*v[2], y=R[3]*v[0]+R[4]*v[1]+R[5]*v[2], z=R[6]*v[0]+R[7]*v[1]+R[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void mulmatTvec3(double* r, const double* R, const double* v) {
  FL(15);
  double x=R[0]*v[0]+R[3]*v[1]+R[6]*v[2], y=R[1]*v[0]+R[4]*v[1]+R[7]*v[2], z=R[2]*v[0]+R[5]*v[1]+R[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void quat2mat(double* R, const double* q) {
  FL(39);
  double w=q[0],x=q[1],y=q[2],z=q[3];
  R[0]=w*w+x*x-y*y-z*z; R[1]=2*(x*y-w*z); R[2]=2*(x*z+w*y);
  R[3]=2*(x*y+w*z); R[4]=w*w-x*x+y*y-z*z; R[5]=2*(y*z-w*x);
  R[6]=2*(x*z-w*y); R[7]=2*(y*z+w*x); R[8]=w*w-x*x-y*y+z*z;
}
static void mulquat(double* r, const double* a, const double* b) {
  FL(28);
  double t[4] = { a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                  a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0] };
  memcpy(r, t, sizeof t);
}
static void normalize4(double* q) {
  FL(13);
  double n = sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]);
  if (n < MINVAL) { q[0]=1; q[1]=q[2]=q[3]=0; } else { q[0]/=n; q[1]/=n; q[2]/=n; q[3]/=n; }
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  FL(7);
  double s = sin(angle*0.5);
  q[0]=cos(angle*0.5); q[1]=axis[0]*s; q[2]=axis[1]*s; q[3]=axis[2]*s;
}

/* ------------------------------------------------------------------ blob parsing */
static const myo_blob_field* blob_find(const void* blob, const char* name) {
  const myo_blob_header* h = (const myo_blob_header*)blob;
  const myo_blob_field* f = (const myo_blob_field*)((const char*)blob + sizeof(myo_blob_header));
  for (uint32_t i = 0; i < h->n_fields; ++i)
    if (strncmp(f[i].name, name, MYO_BLOB_NAME_LEN) == 0) return &f[i];
  return NULL;
}

OrcModel* orc_model_from_blob(const void* src, size_t nbytes, char* err, int errlen) {
  const myo_blob_header* h = (const myo_blob_header*)src;
  if (nbytes < sizeof *h || h->magic != MYO_BLOB_MAGIC || h->version != MYO_BLOB_VERSION ||
      h->total_bytes != nbytes) {
    if (err) snprintf(err, errlen, "bad model blob header");
    return NULL;
  }
  OrcModel* m = (OrcModel*)calloc(1, sizeof *m);
  m->blob = malloc(nbytes);
  memcpy(m->blob, src, nbytes);
  const char* base = (const char*)m->blob;
  int missing = 0;
#define GETI(n) do { const myo_blob_field* f = blob_find(m->blob, #n); \
    if (!f || f->dtype != MYO_BLOB_I32) { if (err) snprintf(err, errlen, "missing int field %s", #n); missing = 1; } \
    else m->n = (const int*)(base + f->offset); } while (0)
#define GETD(n) do { const myo_blob_field* f = blob_find(m->blob, #n); \
    if (!f || f->dtype != MYO_BLOB_F64) { if (err) snprintf(err, errlen, "missing f64 field %s", #n); missing = 1; } \
    else m->n = (const double*)(base + f->offset); } while (0)
  GETI(body_parentid); GETI(body_rootid); GETI(body_weldid); GETI(body_jntnum); GETI(body_jntadr);
  GETI(body_dofnum); GETI(body_dofadr); GETI(jnt_type); GETI(jnt_qposadr); GETI(jnt_dofadr);
  GETI(jnt_bodyid); GETI(jnt_limited); GETI(dof_bodyid); GETI(dof_jntid); GETI(dof_parentid);
  GETI(geom_type); GETI(geom_condim); GETI(geom_bodyid); GETI(geom_priority); GETI(site_bodyid);
  GETI(tendon_adr); GETI(tendon_num); GETI(tendon_limited); GETI(wrap_type); GETI(wrap_objid);
  GETI(actuator_trntype); GETI(actuator_dyntype); GETI(actuator_gaintype); GETI(actuator_biastype);
  GETI(actuator_trnid); GETI(actuator_ctrllimited); GETI(actuator_forcelimited);
  GETD(qpos0); GETD(qpos_spring); GETD(body_pos); GETD(body_quat); GETD(body_ipos); GETD(body_iquat);
  GETD(body_mass); GETD(body_inertia); GETD(body_invweight0); GETD(jnt_solref); GETD(jnt_solimp);
  GETD(jnt_pos); GETD(jnt_axis); GETD(jnt_stiffness); GETD(jnt_range); GETD(jnt_margin);
  GETD(dof_armature); GETD(dof_damping); GETD(dof_invweight0); GETD(geom_solmix); GETD(geom_solref);
  GETD(geom_solimp); GETD(geom_size); GETD(geom_rbound); GETD(geom_pos); GETD(geom_quat);
  GETD(geom_friction); GETD(geom_margin); GETD(geom_gap); GETD(site_pos); GETD(tendon_solref_lim);
  GETD(tendon_solimp_lim); GETD(tendon_range); GETD(tendon_margin); GETD(tendon_stiffness);
  GETD(tendon_damping); GETD(tendon_lengthspring); GETD(tendon_invweight0); GETD(wrap_prm);
  GETD(actuator_dynprm); GETD(actuator_gainprm); GETD(actuator_biasprm); GETD(actuator_ctrlrange);
  GETD(actuator_forcerange); GETD(actuator_gear); GETD(actuator_acc0); GETD(actuator_lengthrange);
#define GETD_OPT(n) do { const myo_blob_field* f = blob_find(m->blob, #n); m->n = (f && f->dtype == MYO_BLOB_F64) ? (const double*)(base + f->offset) : NULL; } while (0)
  GETD_OPT(dof_frictionloss); GETD_OPT(dof_solref); GETD_OPT(dof_solimp);
  GETD_OPT(tendon_frictionloss); GETD_OPT(tendon_solref_fri); GETD_OPT(tendon_solimp_fri);
  const myo_blob_field *fs = blob_find(m->blob, "sizes"), *fi = blob_find(m->blob, "opt_int"),
                       *fd = blob_find(m->blob, "opt_f64"), *p1 = blob_find(m->blob, "x_pair_geom1"),
                       *p2 = blob_find(m->blob, "x_pair_geom2");
  if (missing || !fs || !fi || !fd || !p1 || !p2) {
    if (err && !missing) snprintf(err, errlen, "missing sizes/opt/pair fields");
    free(m->blob); free(m); return NULL;
  }
  const int* s = (const int*)(base + fs->offset);
  m->nq=s[0]; m->nv=s[1]; m->nu=s[2]; m->na=s[3]; m->nbody=s[4]; m->njnt=s[5]; m->ngeom=s[6];
  m->nsite=s[7]; m->ntendon=s[8]; m->nwrap=s[9];
  const int* oi = (const int*)(base + fi->offset);
  m->integrator=oi[0]; m->cone=oi[1]; m->iterations=oi[2]; m->disableflags=oi[3];
  const double* od = (const double*)(base + fd->offset);
  m->timestep=od[0]; m->tolerance=od[1]; m->impratio=od[2];
  m->gravity[0]=od[3]; m->gravity[1]=od[4]; m->gravity[2]=od[5]; m->o_margin=od[6]; m->meaninertia=od[7];
  m->npair = (int)p1->count;
  m->pair_geom1 = (const int*)(base + p1->offset);
  m->pair_geom2 = (const int*)(base + p2->offset);
  { const myo_blob_field* ps = blob_find(m->blob, "x_pair_sub"); m->pair_sub = (ps && ps->count == p1->count) ? (const int*)(base + ps->offset) : NULL; }
  { const myo_blob_field* px = blob_find(m->blob, "x_pair_explicit"); m->pair_explicit = (px && px->dtype == MYO_BLOB_I32 && px->count == p1->count) ? (const int*)(base + px->offset) : NULL;
    const myo_blob_field* xd = blob_find(m->blob, "x_xp_dim"); m->xp_dim = (xd && xd->dtype == MYO_BLOB_I32) ? (const int*)(base + xd->offset) : NULL; }
#define GETD_X(n, field) do { const myo_blob_field* f = blob_find(m->blob, field); m->n = (f && f->dtype == MYO_BLOB_F64) ? (const double*)(base + f->offset) : NULL; } while (0)
  GETD_X(xp_margin, "x_xp_margin"); GETD_X(xp_gap, "x_xp_gap"); GETD_X(xp_solref, "x_xp_solref"); GETD_X(xp_solimp, "x_xp_solimp"); GETD_X(xp_friction, "x_xp_friction");
  if (!m->xp_dim || !m->xp_margin || !m->xp_gap || !m->xp_solref || !m->xp_solimp || !m->xp_friction) m->pair_explicit = NULL;
  return m;
}

void orc_model_free(OrcModel* m) { if (m) { free(m->blob); free(m); } }

int orc_model_int(const OrcModel* m, const char* n) {
#define MI(x) if (!strcmp(n, #x)) return m->x
  MI(nq); MI(nv); MI(nu); MI(na); MI(nbody); MI(njnt); MI(ngeom); MI(nsite); MI(ntendon);
  MI(nwrap); MI(npair); MI(integrator); MI(iterations);
  return -1;
}

/* ------------------------------------------------------------------ data */
static double* falloc(OrcData* d, const char* name, int n) {
  double* p = (double*)calloc(n > 0 ? n : 1, sizeof(double));
  d->fields[d->nfields].name = name; d->fields[d->nfields].p = p; d->fields[d->nfields].n = n;
  d->nfields++;
  return p;
}

OrcData* orc_data_new(const OrcModel* m) {
  OrcData* d = (OrcData*)calloc(1, sizeof *d);
  d->m = m;
  int nv=m->nv, nb=m->nbody;
#define A(name, n) d->name = falloc(d, #name, (n))
  A(qpos, m->nq); A(qvel, nv); A(act, m->na); A(ctrl, m->nu); A(qacc_warmstart, nv);
  A(body_mass, nb); A(geom_friction, 3*m->ngeom); A(geom_size, 3*m->ngeom); A(geom_pos, 3*m->ngeom); A(site_pos, 3*m->nsite);
  A(xpos, 3*nb); A(xquat, 4*nb); A(xmat, 9*nb); A(xipos, 3*nb); A(ximat, 9*nb);
  A(xanchor, 3*m->njnt); A(xaxis, 3*m->njnt); A(geom_xpos, 3*m->ngeom); A(geom_xmat, 9*m->ngeom);
  A(site_xpos, 3*m->nsite); A(subtree_com, 3*nb); A(cinert, 10*nb); A(cdof, 6*nv); A(crb, 10*nb);
  A(ten_length, m->ntendon); A(ten_J, m->ntendon*nv); A(actuator_length, m->nu);
  A(actuator_moment, m->nu*nv); A(M, nv*nv); A(Mchol, nv*nv);
  A(ten_velocity, m->ntendon); A(actuator_velocity, m->nu); A(cvel, 6*nb); A(cdof_dot, 6*nv);
  A(qfrc_passive, nv); A(qfrc_bias, nv);
  A(act_dot, m->na); A(actuator_force, m->nu); A(qfrc_actuator, nv); A(qfrc_smooth, nv);
  A(qacc_smooth, nv); A(qacc, nv); A(qfrc_constraint, nv);
  A(efc_J, MAXEFC*nv); A(efc_pos, MAXEFC); A(efc_margin, MAXEFC); A(efc_D, MAXEFC); A(efc_R, MAXEFC);
  A(efc_aref, MAXEFC); A(efc_vel, MAXEFC); A(efc_force, MAXEFC); A(efc_diagApprox, MAXEFC); A(efc_floss, MAXEFC);
  A(efc_KBIP, 4*MAXEFC);
  A(w1, MAXEFC+nv); A(w2, MAXEFC+nv); A(w3, MAXEFC+nv); A(w4, MAXEFC+nv); A(w5, MAXEFC+nv);
  A(H, nv*nv); A(cacc, 6*nb); A(cfrc, 6*nb);
#undef A
  orc_reset(m, d);
  return d;
}

void orc_data_free(OrcData* d) {
  if (!d) return;
  for (int i = 0; i < d->nfields; ++i) free(d->fields[i].p);
  free(d);
}

void orc_reset(const OrcModel* m, OrcData* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double)*m->nq);
  memset(d->qvel, 0, sizeof(double)*m->nv);
  memset(d->act, 0, sizeof(double)*m->na);
  memset(d->ctrl, 0, sizeof(double)*m->nu);
  memset(d->qacc_warmstart, 0, sizeof(double)*m->nv);
  memcpy(d->body_mass, m->body_mass, sizeof(double)*m->nbody);
  memcpy(d->geom_friction, m->geom_friction, sizeof(double)*3*m->ngeom);
  memcpy(d->geom_size, m->geom_size, sizeof(double)*3*m->ngeom);
  memcpy(d->geom_pos, m->geom_pos, sizeof(double)*3*m->ngeom);
  memcpy(d->site_pos, m->site_pos, sizeof(double)*3*m->nsite);
  d->time = 0; d->bad = 0;
}

double* orc_ptr(OrcData* d, const char* name) {
  if (!strcmp(name, "time")) return &d->time;
  for (int i = 0; i < d->nfields; ++i) if (!strcmp(d->fields[i].name, name)) return d->fields[i].p;
  return NULL;
}
int orc_count(const OrcData* d, const char* name) {
  for (int i = 0; i < d->nfields; ++i) if (!strcmp(d->fields[i].name, name)) return d->fields[i].n;
  return -1;
}
int orc_get_int(const OrcData* d, const char* n) {
  if (!strcmp(n, "ncon")) return d->ncon;
  if (!strcmp(n, "nefc")) return d->nefc;
  if (!strcmp(n, "solver_iter")) return d->solver_iter;
  if (!strcmp(n, "bad")) return d->bad;
  if (!strcmp(n, "nl")) return d->nl;
  if (!strcmp(n, "ntl")) return d->ntl;
  return -1;
}

/* ------------------------------------------------------------------ P2 kinematics */
void orc_kinematics(const OrcModel* m, OrcData* d) {
  double* xpos=d->xpos; double* xquat=d->xquat; double* xmat=d->xmat;
  xpos[0]=xpos[1]=xpos[2]=0; xquat[0]=1; xquat[1]=xquat[2]=xquat[3]=0; quat2mat(xmat, xquat);
  memcpy(d->xipos, xpos, 3*sizeof(double)); memcpy(d->ximat, xmat, 9*sizeof(double));
  for (int b = 1; b < m->nbody; ++b) {
    int par = m->body_parentid[b];
    int jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    double p[3], q[4];
    if (jn == 1 && m->jnt_type[ja] == MYO_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      normalize4(d->qpos + qa + 3);
      memcpy(p, d->qpos + qa, 3*sizeof(double)); memcpy(q, d->qpos + qa + 3, 4*sizeof(double));
      memcpy(d->xanchor + 3*ja, p, 3*sizeof(double));
      d->xaxis[3*ja]=0; d->xaxis[3*ja+1]=0; d->xaxis[3*ja+2]=1;
    } else {
      double t[3];
      mulmatvec3(t, xmat + 9*par, m->body_pos + 3*b);
      p[0]=xpos[3*par]+t[0]; p[1]=xpos[3*par+1]+t[1]; p[2]=xpos[3*par+2]+t[2];
      mulquat(q, xquat + 4*par, m->body_quat + 4*b);
      for (int k = 0; k < jn; ++k) {
        int j = ja + k, qa = m->jnt_qposadr[j];
        double R[9], anchor[3], axis[3];
        quat2mat(R, q);
        mulmatvec3(anchor, R, m->jnt_pos + 3*j); anchor[0]+=p[0]; anchor[1]+=p[1]; anchor[2]+=p[2];
        mulmatvec3(axis, R, m->jnt_axis + 3*j);
        memcpy(d->xanchor + 3*j, anchor, sizeof anchor); memcpy(d->xaxis + 3*j, axis, sizeof axis);
        double ang = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == MYO_JNT_SLIDE) {
          p[0]+=axis[0]*ang; p[1]+=axis[1]*ang; p[2]+=axis[2]*ang;
        } else { /* hinge: rotate about the local axis, keep the anchor fixed */
          double qloc[4], R2[9], t2[3];
          axisangle2quat(qloc, m->jnt_axis + 3*j, ang);
          mulquat(q, q, qloc);
          quat2mat(R2, q);
          mulmatvec3(t2, R2, m->jnt_pos + 3*j);
          p[0]=anchor[0]-t2[0]; p[1]=anchor[1]-t2[1]; p[2]=anchor[2]-t2[2];
        }
      }
    }
    normalize4(q);
    memcpy(xpos + 3*b, p, sizeof p); memcpy(xquat + 4*b, q, sizeof q);
    quat2mat(xmat + 9*b, q);
    double t[3], qi[4];
    mulmatvec3(t, xmat + 9*b, m->body_ipos + 3*b);
    d->xipos[3*b]=p[0]+t[0]; d->xipos[3*b+1]=p[1]+t[1]; d->xipos[3*b+2]=p[2]+t[2];
    mulquat(qi, q, m->body_iquat + 4*b);
    quat2mat(d->ximat + 9*b, qi);
  }
  for (int g = 0; g < m->ngeom; ++g) {
    int b = m->geom_bodyid[g]; double t[3], q[4];
    mulmatvec3(t, xmat + 9*b, d->geom_pos + 3*g);   /* per-data copy: the reorient reset rewrites the die's geom_pos */
    for (int k = 0; k < 3; ++k) d->geom_xpos[3*g+k] = xpos[3*b+k] + t[k];
    mulquat(q, xquat + 4*b, m->geom_quat + 4*g);
    quat2mat(d->geom_xmat + 9*g, q);
  }
  for (int s = 0; s < m->nsite; ++s) {
    int b = m->site_bodyid[s]; double t[3];
    mulmatvec3(t, xmat + 9*b, d->site_pos + 3*s);
    for (int k = 0; k < 3; ++k) d->site_xpos[3*s+k] = xpos[3*b+k] + t[k];
  }
}

/* ------------------------------------------------------------------ P2 comPos (cinert, cdof) */
static void com_pos(const OrcModel* m, OrcData* d) {
  int nb = m->nbody;
  FL(nb*(3 + 4 + 3) + nb*(27 + 45 + 30) + m->nv*12);    /* subtree com sums; inertia rotated into the world and shifted (cinert); cdof */
  double* sc = d->subtree_com; double* mass = d->w1; /* subtree mass scratch */
  for (int b = 0; b < nb; ++b) {
    mass[b] = d->body_mass[b];
    for (int k = 0; k < 3; ++k) sc[3*b+k] = d->body_mass[b]*d->xipos[3*b+k];
  }
  for (int b = nb-1; b > 0; --b) {
    int p = m->body_parentid[b];
    mass[p] += mass[b];
    for (int k = 0; k < 3; ++k) sc[3*p+k] += sc[3*b+k];
  }
  for (int b = 0; b < nb; ++b) {
    if (mass[b] < MINVAL) for (int k = 0; k < 3; ++k) sc[3*b+k] = d->xipos[3*b+k];
    else for (int k = 0; k < 3; ++k) sc[3*b+k] /= mass[b];
  }
  for (int b = 1; b < nb; ++b) {
    const double* R = d->ximat + 9*b; const double* I = m->body_inertia + 3*b;
    const double* c = sc + 3*m->body_rootid[b];
    double off[3] = { d->xipos[3*b]-c[0], d->xipos[3*b+1]-c[1], d->xipos[3*b+2]-c[2] };
    double mb = d->body_mass[b];
    double Iw[6]; /* xx yy zz xy xz yz */
    Iw[0]=R[0]*R[0]*I[0]+R[1]*R[1]*I[1]+R[2]*R[2]*I[2];
    Iw[1]=R[3]*R[3]*I[0]+R[4]*R[4]*I[1]+R[5]*R[5]*I[2];
    Iw[2]=R[6]*R[6]*I[0]+R[7]*R[7]*I[1]+R[8]*R[8]*I[2];
    Iw[3]=R[0]*R[3]*I[0]+R[1]*R[4]*I[1]+R[2]*R[5]*I[2];
    Iw[4]=R[0]*R[6]*I[0]+R[1]*R[7]*I[1]+R[2]*R[8]*I[2];
    Iw[5]=R[3]*R[6]*I[0]+R[4]*R[7]*I[1]+R[5]*R[8]*I[2];
    double* ci = d->cinert + 10*b;
    ci[0]=Iw[0]+mb*(off[1]*off[1]+off[2]*off[2]);
    ci[1]=Iw[1]+mb*(off[0]*off[0]+off[2]*off[2]);
    ci[2]=Iw[2]+mb*(off[0]*off[0]+off[1]*off[1]);
    ci[3]=Iw[3]-mb*off[0]*off[1]; ci[4]=Iw[4]-mb*off[0]*off[2]; ci[5]=Iw[5]-mb*off[1]*off[2];
    ci[6]=mb*off[0]; ci[7]=mb*off[1]; ci[8]=mb*off[2]; ci[9]=mb;
  }
  memset(d->cinert, 0, 10*sizeof(double));
  for (int j = 0; j < m->njnt; ++j) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* c = sc + 3*m->body_rootid[b];
    double off[3] = { c[0]-d->xanchor[3*j], c[1]-d->xanchor[3*j+1], c[2]-d->xanchor[3*j+2] };
    if (m->jnt_type[j] == MYO_JNT_FREE) {
      for (int k = 0; k < 3; ++k) {
        double* cd = d->cdof + 6*(da+k); memset(cd, 0, 6*sizeof(double)); cd[3+k] = 1;
      }
      for (int k = 0; k < 3; ++k) {
        double* cd = d->cdof + 6*(da+3+k);
        double ax[3] = { d->xmat[9*b+k], d->xmat[9*b+3+k], d->xmat[9*b+6+k] };
        memcpy(cd, ax, sizeof ax); cross3(cd+3, ax, off);
      }
    } else if (m->jnt_type[j] == MYO_JNT_SLIDE) {
      double* cd = d->cdof + 6*da; cd[0]=cd[1]=cd[2]=0; memcpy(cd+3, d->xaxis + 3*j, 3*sizeof(double));
    } else {
      double* cd = d->cdof + 6*da; memcpy(cd, d->xaxis + 3*j, 3*sizeof(double)); cross3(cd+3, d->xaxis + 3*j, off);
    }
  }
}

/* spatial inertia (10-vector about the tree reference point) times motion vector */
static void mul_inert_vec(double* r, const double* I, const double* v) {
  const double* w = v; const double* l = v+3; const double* h = I+6; double mass = I[9];
  double t[3];
  FL(15 + 3 + 9);
  r[0]=I[0]*w[0]+I[3]*w[1]+I[4]*w[2]; r[1]=I[3]*w[0]+I[1]*w[1]+I[5]*w[2]; r[2]=I[4]*w[0]+I[5]*w[1]+I[2]*w[2];
  cross3(t, h, l); r[0]+=t[0]; r[1]+=t[1]; r[2]+=t[2];
  cross3(t, h, w); r[3]=mass*l[0]-t[0]; r[4]=mass*l[1]-t[1]; r[5]=mass*l[2]-t[2];
}
static void cross_motion(double* r, const double* v, const double* mvec) {
  double a[3], b[3], c[3];
  cross3(a, v, mvec); cross3(b, v, mvec+3); cross3(c, v+3, mvec); FL(3);
  r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; r[3]=b[0]+c[0]; r[4]=b[1]+c[1]; r[5]=b[2]+c[2];
}
static void cross_force(double* r, const double* v, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v+3, f+3); cross3(c, v, f+3); FL(3);
  r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; r[3]=c[0]; r[4]=c[1]; r[5]=c[2];
}

/* ------------------------------------------------------------------ P5 CRB + factor */
static int chol_factor(double* L, const double* A, int n) {
  if (L != A) memcpy(L, A, sizeof(double)*n*n);
  for (int k = 0; k < n; ++k) {
    double s = L[k*n+k];
    for (int j = 0; j < k; ++j) s -= L[k*n+j]*L[k*n+j];
    FL(2*k + 1 + (n-k-1)*(2*k+1));
    if (s < MINVAL) s = MINVAL;
    double dk = sqrt(s); L[k*n+k] = dk;
    for (int i = k+1; i < n; ++i) {
      double t = L[i*n+k];
      for (int j = 0; j < k; ++j) t -= L[i*n+j]*L[k*n+j];
      L[i*n+k] = t/dk;
    }
  }
  return 0;
}
static void chol_solve(const double* L, double* x, int n) { /* in place */
  FL(2*n*n);
  for (int i = 0; i < n; ++i) { double s = x[i]; for (int j = 0; j < i; ++j) s -= L[i*n+j]*x[j]; x[i] = s/L[i*n+i]; }
  for (int i = n-1; i >= 0; --i) { double s = x[i]; for (int j = i+1; j < n; ++j) s -= L[j*n+i]*x[j]; x[i] = s/L[i*n+i]; }
}

static void crb(const OrcModel* m, OrcData* d) {
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, sizeof(double)*10*nb);
  for (int b = nb-1; b > 0; --b) {
    int p = m->body_parentid[b];
    if (p > 0) { for (int k = 0; k < 10; ++k) d->crb[10*p+k] += d->crb[10*b+k]; FL(10); }
  }
  memset(d->M, 0, sizeof(double)*nv*nv);
  for (int i = 0; i < nv; ++i) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10*m->dof_bodyid[i], d->cdof + 6*i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double s = 0; for (int k = 0; k < 6; ++k) s += d->cdof[6*j+k]*buf[k];
      FL(12);
      d->M[i*nv+j] = s; d->M[j*nv+i] = s;
    }
    d->M[i*nv+i] += m->dof_armature[i];
  }
  chol_factor(d->Mchol, d->M, nv);
}

/* ------------------------------------------------------------------ Jacobian of a point */
static void jac_point(const OrcModel* m, const OrcData* d, int body, const double* point,
                      double* jacp /* 3 x nv */) {
  int nv = m->nv;
  memset(jacp, 0, sizeof(double)*3*nv);
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  const double* c = d->subtree_com + 3*m->body_rootid[body];
  double off[3] = { point[0]-c[0], point[1]-c[1], point[2]-c[2] }; FL(3);
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    const double* cd = d->cdof + 6*i; double t[3];
    cross3(t, cd, off); FL(3);
    jacp[i] = cd[3]+t[0]; jacp[nv+i] = cd[4]+t[1]; jacp[2*nv+i] = cd[5]+t[2];
    i = m->dof_parentid[i];
  }
}

/* ------------------------------------------------------------------ P3 tendon wrapping */
static int is_intersect(const double* p1, const double* p2, const double* p3, const double* p4) {
  double det = (p4[1]-p3[1])*(p2[0]-p1[0]) - (p4[0]-p3[0])*(p2[1]-p1[1]);
  if (fabs(det) < MINVAL) return 0;
  double a = ((p4[0]-p3[0])*(p1[1]-p3[1]) - (p4[1]-p3[1])*(p1[0]-p3[0]))/det;
  double b = ((p2[0]-p1[0])*(p1[1]-p3[1]) - (p2[1]-p1[1])*(p1[0]-p3[0]))/det;
  return (a >= 0 && a <= 1 && b >= 0 && b <= 1);
}

/* 2-D circle wrap: end points d[0:2], d[2:4], optional side point sd, radius rad.
 * Returns arc length (>=0) and the two tangent points in pnt, or -1 for a straight path. */
static double wrap_circle(double* pnt, const double* dd, const double* sd, double rad) {
  double sqlen0 = dd[0]*dd[0]+dd[1]*dd[1], sqlen1 = dd[2]*dd[2]+dd[3]*dd[3], sqrad = rad*rad;
  double dif[2] = { dd[2]-dd[0], dd[3]-dd[1] };
  double dsq = dif[0]*dif[0]+dif[1]*dif[1];
  FL(12);
  if (sqlen0 < sqrad || sqlen1 < sqrad || rad < MINVAL) return -1;
  if (dsq < MINVAL) return -1;
  double a = -(dif[0]*dd[0]+dif[1]*dd[1])/dsq;
  if (a < 0) a = 0; else if (a > 1) a = 1;
  double tmp[2] = { a*dif[0]+dd[0], a*dif[1]+dd[1] };
  FL(5 + 4 + 6);
  if (tmp[0]*tmp[0]+tmp[1]*tmp[1] > sqrad && (!sd || sd[0]*tmp[0]+sd[1]*tmp[1] >= 0)) return -1;
  double sol[2][4], good[2];
  double sqrt0 = sqrt(sqlen0 - sqrad), sqrt1 = sqrt(sqlen1 - sqrad);
  FL(4 + 2*(4*6 + 10) + 2*2*12 + 6);      /* two candidate tangent pairs, their side / length score, two segment-crossing tests each, the arc */
  for (int i = 0; i < 2; ++i) {
    int sgn = (i == 0 ? 1 : -1);
    sol[i][0] = (dd[0]*sqrad + sgn*rad*dd[1]*sqrt0)/sqlen0;
    sol[i][1] = (dd[1]*sqrad - sgn*rad*dd[0]*sqrt0)/sqlen0;
    sol[i][2] = (dd[2]*sqrad - sgn*rad*dd[3]*sqrt1)/sqlen1;
    sol[i][3] = (dd[3]*sqrad + sgn*rad*dd[2]*sqrt1)/sqlen1;
    if (sd) {
      double t[2] = { sol[i][0]+sol[i][2], sol[i][1]+sol[i][3] };
      double n = sqrt(t[0]*t[0]+t[1]*t[1]);
      if (n < MINVAL) { t[0]=1; t[1]=0; } else { t[0]/=n; t[1]/=n; }
      good[i] = t[0]*sd[0]+t[1]*sd[1];
    } else {
      double t[2] = { sol[i][0]-sol[i][2], sol[i][1]-sol[i][3] };
      good[i] = -(t[0]*t[0]+t[1]*t[1]);
    }
    if (is_intersect(dd, sol[i], dd+2, sol[i]+2)) good[i] = -10000;
  }
  int i = (good[0] > good[1]) ? 0 : 1;
  memcpy(pnt, sol[i], 4*sizeof(double));
  if (is_intersect(dd, pnt, dd+2, pnt+2)) return -1;
  double c = (pnt[0]*pnt[2]+pnt[1]*pnt[3])/sqrad;
  if (c > 1) c = 1; else if (c < -1) c = -1;
  return rad*acos(c);
}

static double wrap_geom(double* wpnt /*6*/, const double* x0, const double* x1, const double* xpos,
                        const double* xmat, double radius, int type, const double* side) {
  double p0[3], p1[3], t[3];
  FL(6);
  for (int k = 0; k < 3; ++k) t[k] = x0[k]-xpos[k];
  mulmatTvec3(p0, xmat, t);
  for (int k = 0; k < 3; ++k) t[k] = x1[k]-xpos[k];
  mulmatTvec3(p1, xmat, t);
  if (norm3(p0) < MINVAL || norm3(p1) < MINVAL) return -1;
  double axis0[3], axis1[3];
  if (type == MYO_WRAP_SPHERE) {
    double normal[3];
    memcpy(axis0, p0, sizeof p0); normalize3(axis0);
    cross3(normal, p0, p1);
    double nrm = norm3(normal);
    if (nrm < MINVAL) { /* p0, p1 and the centre collinear: any plane through them */
      int imin = 0; if (fabs(axis0[1]) < fabs(axis0[imin])) imin = 1; if (fabs(axis0[2]) < fabs(axis0[imin])) imin = 2;
      double e[3] = {0,0,0}; e[imin] = 1;
      cross3(normal, axis0, e); normalize3(normal);
    } else { normal[0]/=nrm; normal[1]/=nrm; normal[2]/=nrm; }
    cross3(axis1, normal, axis0); normalize3(axis1);
  } else {
    axis0[0]=1; axis0[1]=0; axis0[2]=0; axis1[0]=0; axis1[1]=1; axis1[2]=0;
  }
  double s[4] = { dot3(p0,axis0), dot3(p0,axis1), dot3(p1,axis0), dot3(p1,axis1) };
  double sd[2]; int has_side = 0;
  if (side) {
    double ps[3];
    for (int k = 0; k < 3; ++k) t[k] = side[k]-xpos[k];
    mulmatTvec3(ps, xmat, t);
    sd[0] = dot3(ps,axis0); sd[1] = dot3(ps,axis1);
    double n = sqrt(sd[0]*sd[0]+sd[1]*sd[1]);
    if (n < MINVAL) { sd[0]=1; sd[1]=0; } else { sd[0]/=n; sd[1]/=n; }
    sd[0]*=radius; sd[1]*=radius;
    has_side = 1;
  }
  double pnt[4];
  double wlen = wrap_circle(pnt, s, has_side ? sd : NULL, radius);
  if (wlen < 0) return -1;
  double r0[3], r1[3];
  FL(18 + 6 + (type == MYO_WRAP_CYLINDER ? 30 : 0));
  for (int k = 0; k < 3; ++k) { r0[k] = axis0[k]*pnt[0]+axis1[k]*pnt[1]; r1[k] = axis0[k]*pnt[2]+axis1[k]*pnt[3]; }
  if (type == MYO_WRAP_CYLINDER) {
    double L0 = sqrt((s[0]-pnt[0])*(s[0]-pnt[0])+(s[1]-pnt[1])*(s[1]-pnt[1]));
    double L1 = sqrt((s[2]-pnt[2])*(s[2]-pnt[2])+(s[3]-pnt[3])*(s[3]-pnt[3]));
    r0[2] = p0[2] + (p1[2]-p0[2])*L0/(L0+wlen+L1);
    r1[2] = p0[2] + (p1[2]-p0[2])*(L0+wlen)/(L0+wlen+L1);
    double height = fabs(r1[2]-r0[2]);
    wlen = sqrt(wlen*wlen + height*height);
  }
  mulmatvec3(wpnt, xmat, r0); mulmatvec3(wpnt+3, xmat, r1);
  for (int k = 0; k < 3; ++k) { wpnt[k] += xpos[k]; wpnt[3+k] += xpos[k]; }
  return wlen;
}

static void tendon(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  double* jac0 = d->w1; /* reuse big scratch: need 3*nv each */
  double* jac1 = d->w2;
  memset(d->ten_J, 0, sizeof(double)*m->ntendon*nv);
  for (int t = 0; t < m->ntendon; ++t) {
    int adr = m->tendon_adr[t], num = m->tendon_num[t];
    double len = 0, divisor = 1; double* J = d->ten_J + t*nv;
    int j = 0;
    while (j < num-1) {
      int type0 = m->wrap_type[adr+j], type1 = m->wrap_type[adr+j+1];
      int id0 = m->wrap_objid[adr+j], id1 = m->wrap_objid[adr+j+1];
      if (type0 == MYO_WRAP_PULLEY || type1 == MYO_WRAP_PULLEY) {
        if (type0 == MYO_WRAP_PULLEY) divisor = m->wrap_prm[adr+j];
        j++; continue;
      }
      double wpnt[12]; int wbody[4]; int wcnt; double wlen = -1; int idg = -1;
      wbody[0] = m->site_bodyid[id0]; memcpy(wpnt, d->site_xpos + 3*id0, 3*sizeof(double));
      if (type1 == MYO_WRAP_SPHERE || type1 == MYO_WRAP_CYLINDER) {
        idg = id1; id1 = m->wrap_objid[adr+j+2];
        int sideid = (int)lround(m->wrap_prm[adr+j+1]);
        const double* side = sideid >= 0 ? d->site_xpos + 3*sideid : NULL;
        wlen = wrap_geom(wpnt+3, d->site_xpos + 3*id0, d->site_xpos + 3*id1, d->geom_xpos + 3*idg,
                         d->geom_xmat + 9*idg, d->geom_size[3*idg], type1, side);
      }
      if (wlen < 0) {
        wbody[1] = m->site_bodyid[id1]; memcpy(wpnt+3, d->site_xpos + 3*id1, 3*sizeof(double)); wcnt = 2;
      } else {
        wbody[1] = wbody[2] = m->geom_bodyid[idg]; wbody[3] = m->site_bodyid[id1];
        memcpy(wpnt+9, d->site_xpos + 3*id1, 3*sizeof(double)); wcnt = 4;
      }
      for (int k = 0; k < wcnt-1; ++k) {
        if (wcnt == 4 && k == 1) { len += wlen/divisor; continue; }
        double dif[3] = { wpnt[3*k+3]-wpnt[3*k], wpnt[3*k+4]-wpnt[3*k+1], wpnt[3*k+5]-wpnt[3*k+2] };
        double dn = norm3(dif); FL(3 + 2);
        len += dn/divisor;
        if (wbody[k] != wbody[k+1] && dn > MINVAL) {
          dif[0]/=dn; dif[1]/=dn; dif[2]/=dn;
          jac_point(m, d, wbody[k], wpnt+3*k, jac0);
          jac_point(m, d, wbody[k+1], wpnt+3*k+3, jac1);
          for (int c = 0; c < nv; ++c) {
            if (jac1[c] != jac0[c] || jac1[nv+c] != jac0[nv+c] || jac1[2*nv+c] != jac0[2*nv+c]) FL(10);   /* structurally non-zero columns only */
            J[c] += (dif[0]*(jac1[c]-jac0[c]) + dif[1]*(jac1[nv+c]-jac0[nv+c]) + dif[2]*(jac1[2*nv+c]-jac0[2*nv+c]))/divisor;
          }
        }
      }
      j += (idg >= 0 ? 2 : 1);
    }
    d->ten_length[t] = len;
  }
}

/* ------------------------------------------------------------------ P4 transmission */
static void transmission(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  memset(d->actuator_moment, 0, sizeof(double)*m->nu*nv);
  for (int i = 0; i < m->nu; ++i) {
    double gear = m->actuator_gear[6*i]; int id = m->actuator_trnid[2*i];
    if (m->actuator_trntype[i] == MYO_TRN_TENDON) {
      d->actuator_length[i] = gear*d->ten_length[id];
      for (int c = 0; c < nv; ++c) { d->actuator_moment[i*nv+c] = gear*d->ten_J[id*nv+c]; if (d->ten_J[id*nv+c] != 0) FL(1); }
    } else { /* joint (hinge/slide) */
      d->actuator_length[i] = gear*d->qpos[m->jnt_qposadr[id]];
      d->actuator_moment[i*nv + m->jnt_dofadr[id]] = gear;
    }
  }
}

/* ------------------------------------------------------------------ P6 collision */
static void make_frame(double* f) {
  normalize3(f);
  f[3]=f[4]=f[5]=0;
  if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  double t = dot3(f, f+3);
  f[3]-=t*f[0]; f[4]-=t*f[1]; f[5]-=t*f[2];
  normalize3(f+3);
  cross3(f+6, f, f+3);
}

static int sphere_sphere_raw(double* dist, double* pos, double* n, const double* c1, double r1,
                             const double* c2, double r2, double margin) {
  double dif[3] = { c2[0]-c1[0], c2[1]-c1[1], c2[2]-c1[2] };
  double cd = norm3(dif);
  if (cd - r1 - r2 > margin) return 0;
  if (cd < MINVAL) { n[0]=1; n[1]=0; n[2]=0; } else { n[0]=dif[0]/cd; n[1]=dif[1]/cd; n[2]=dif[2]/cd; }
  *dist = cd - r1 - r2;
  for (int k = 0; k < 3; ++k) pos[k] = c1[k] + n[k]*(r1 + 0.5*(*dist));
  return 1;
}

static void seg_nearest(double* out, const double* c, const double* axis, double half, const double* p) {
  double t = (p[0]-c[0])*axis[0]+(p[1]-c[1])*axis[1]+(p[2]-c[2])*axis[2];
  if (t > half) t = half; else if (t < -half) t = -half;
  for (int k = 0; k < 3; ++k) out[k] = c[k] + t*axis[k];
}

/* ---- narrow phases beyond MuJoCo's sphere / capsule primitives.  Frames: p = geom centre, R = geom rotation (row-major,
 * columns = the geom's axes in world coordinates), normal always from geom 1 towards geom 2, pos = the point midway between
 * the two surfaces, dist < 0 = penetration (mjContact conventions).
 * [3P-RECALL] MuJoCo 2.1 sends sphere / capsule vs cylinder / ellipsoid through its general convex routine (libccd MPR,
 * tolerance 1e-6) and has primitives mjc_CapsuleBox / mjc_BoxBox / mjc_PlaneCylinder whose multi-contact heuristics are not
 * restated here: what follows is the exact geometry (closest points) with this stepper's own choice of contact points where
 * MuJoCo returns several.  Differences from MuJoCo are therefore possible in the number and placement of contacts of a
 * face-on-face configuration; a single-point configuration agrees with MPR to its tolerance. */
/* d/dt of the signed distance from the point x = c + t a to a solid, up to a positive factor: the minimiser over the segment is
 * where it changes sign (the signed distance to a convex solid is convex along a line), found by bisection to the last bit — a
 * search on the VALUES of the distance only resolves t to sqrt(eps), which showed up as a 6e-9 oracle / device difference */
static double dsd_box(const double* c, const double* a, double t, const double* s) {
  double x[3] = { c[0]+t*a[0], c[1]+t*a[1], c[2]+t*a[2] }, g = 0.0;
  int inside = 1;
  for (int k = 0; k < 3; ++k) {
    if (x[k] > s[k]) { g += (x[k]-s[k])*a[k]; inside = 0; } else if (x[k] < -s[k]) { g += (x[k]+s[k])*a[k]; inside = 0; }
  }
  if (!inside) return g;
  int kb = 0; double best = 1e300;
  for (int k = 0; k < 3; ++k) { double e = s[k]-fabs(x[k]); if (e < best) { best = e; kb = k; } }
  return x[kb] > 0 ? a[kb] : -a[kb];
}
static double dsd_cylinder(const double* c, const double* a, double t, double R, double h) {
  double x[3] = { c[0]+t*a[0], c[1]+t*a[1], c[2]+t*a[2] };
  double rho = sqrt(x[0]*x[0] + x[1]*x[1]), az = fabs(x[2]);
  double ux = rho > MINVAL ? x[0]/rho : 1.0, uy = rho > MINVAL ? x[1]/rho : 0.0;
  if (rho > R || az > h) {
    double qr = rho < R ? rho : R, qz = x[2] > h ? h : (x[2] < -h ? -h : x[2]);
    return (x[0]-ux*qr)*a[0] + (x[1]-uy*qr)*a[1] + (x[2]-qz)*a[2];
  }
  if (h - az < R - rho) return x[2] > 0 ? a[2] : -a[2];
  return ux*a[0] + uy*a[1];
}
#define SEG_ARGMIN(tout, h, DSD_EXPR) { double lo_ = -(h), hi_ = (h), glo_, ghi_;                     \
    { double tt = lo_; glo_ = (DSD_EXPR); } { double tt = hi_; ghi_ = (DSD_EXPR); }                     \
    if (glo_ >= 0) (tout) = lo_; else if (ghi_ <= 0) (tout) = hi_; else {                               \
      for (int it_ = 0; it_ < 60; ++it_) { double tt = 0.5*(lo_ + hi_), g_ = (DSD_EXPR); if (g_ > 0) hi_ = tt; else lo_ = tt; } \
      (tout) = 0.5*(lo_ + hi_); } }

/* sphere (centre c, radius r) against a box: shared by sphere-box, capsule-box and the box-box vertex contacts */
static int point_box(double* dist, double* pos, double* nrm, const double* c, double r, const double* pb, const double* Rb,
                     const double* sb, double margin, int smooth_inside) {
  double t[3] = { c[0]-pb[0], c[1]-pb[1], c[2]-pb[2] }, x[3], cl[3];
  mulmatTvec3(x, Rb, t);
  int inside = 1;
  for (int k = 0; k < 3; ++k) {
    cl[k] = x[k];
    if (cl[k] > sb[k]) { cl[k] = sb[k]; inside = 0; } else if (cl[k] < -sb[k]) { cl[k] = -sb[k]; inside = 0; }
  }
  double nl[3], dd;
  if (!inside) {
    double df[3] = { cl[0]-x[0], cl[1]-x[1], cl[2]-x[2] };
    double dn = norm3(df);
    dd = dn - r;
    if (dd > margin) return 0;
    nl[0]=df[0]/dn; nl[1]=df[1]/dn; nl[2]=df[2]/dn;
  } else { /* centre inside the box: push out through the nearest face */
    int kb = 0; double best = 1e300;
    for (int k = 0; k < 3; ++k) { double e = sb[k]-fabs(x[k]); if (e < best) { best = e; kb = k; } }
    nl[0]=nl[1]=nl[2]=0; nl[kb] = x[kb] > 0 ? -1.0 : 1.0;
    if (smooth_inside) {
      /* a capsule whose AXIS runs inside the box: the nearest point of the segment then sits where two faces are equally near (a
       * ridge of the distance field), where "the nearest face" flips with the last bit — the normal is taken from the smooth
       * field x_k / s_k^2 instead (equal to the face normal over the middle of a face); depth = distance to the nearest face */
      double g[3] = { -x[0]/(sb[0]*sb[0]), -x[1]/(sb[1]*sb[1]), -x[2]/(sb[2]*sb[2]) }, gn = norm3(g);
      if (gn > MINVAL) { nl[0] = g[0]/gn; nl[1] = g[1]/gn; nl[2] = g[2]/gn; }
    }
    dd = -best - r;
    if (dd > margin) return 0;
  }
  mulmatvec3(nrm, Rb, nl);
  *dist = dd;
  for (int k = 0; k < 3; ++k) pos[k] = c[k] + nrm[k]*(r + 0.5*dd);
  return 1;
}
static int point_cylinder(double* dist, double* pos, double* nrm, const double* c, double r, const double* pc, const double* Rc,
                          double R, double h, double margin) {
  double t[3] = { c[0]-pc[0], c[1]-pc[1], c[2]-pc[2] }, x[3], nl[3], dd;
  mulmatTvec3(x, Rc, t);
  double rho = sqrt(x[0]*x[0] + x[1]*x[1]), az = fabs(x[2]);
  double ux = rho > MINVAL ? x[0]/rho : 1.0, uy = rho > MINVAL ? x[1]/rho : 0.0;
  if (rho > R || az > h) {                 /* outside: closest point of the solid */
    double qr = rho < R ? rho : R, qz = x[2] > h ? h : (x[2] < -h ? -h : x[2]);
    double df[3] = { ux*qr - x[0], uy*qr - x[1], qz - x[2] };
    double dn = norm3(df);
    dd = dn - r;
    if (dd > margin) return 0;
    nl[0]=df[0]/dn; nl[1]=df[1]/dn; nl[2]=df[2]/dn;
  } else {                                 /* centre inside: out through the nearer of side wall / cap */
    double e_side = R - rho, e_cap = h - az;
    if (e_cap < e_side) { nl[0]=0; nl[1]=0; nl[2] = x[2] > 0 ? -1.0 : 1.0; dd = -e_cap - r; }
    else { nl[0] = -ux; nl[1] = -uy; nl[2] = 0; dd = -e_side - r; }
    if (dd > margin) return 0;
  }
  mulmatvec3(nrm, Rc, nl);
  *dist = dd;
  for (int k = 0; k < 3; ++k) pos[k] = c[k] + nrm[k]*(r + 0.5*dd);
  return 1;
}
static int point_ellipsoid(double* dist, double* pos, double* nrm, const double* c, double r, const double* pe, const double* Re,
                           const double* s, double margin) {
  double t[3] = { c[0]-pe[0], c[1]-pe[1], c[2]-pe[2] }, x[3], q[3], g[3];
  mulmatTvec3(x, Re, t);
  double lev = (x[0]/s[0])*(x[0]/s[0]) + (x[1]/s[1])*(x[1]/s[1]) + (x[2]/s[2])*(x[2]/s[2]);
  double dn, sign;
  if (lev > 1.0) {
    /* closest surface point: q_k = s_k^2 x_k / (tt + s_k^2) with tt >= 0 the root of sum (s_k x_k / (tt + s_k^2))^2 = 1; bisection,
     * 64 halvings of [0, |x| max(s)] (F is decreasing in tt; F(0) = lev - 1 > 0) */
    double smax = s[0] > s[1] ? s[0] : s[1]; if (s[2] > smax) smax = s[2];
    double lo = 0.0, hi = norm3(x)*smax;
    for (int it = 0; it < 64; ++it) {
      double mid = 0.5*(lo + hi), F = -1.0;
      for (int k = 0; k < 3; ++k) { double v = s[k]*x[k]/(mid + s[k]*s[k]); F += v*v; }
      if (F > 0) lo = mid; else hi = mid;
    }
    double tt = 0.5*(lo + hi);
    for (int k = 0; k < 3; ++k) q[k] = s[k]*s[k]*x[k]/(tt + s[k]*s[k]);
    double df[3] = { q[0]-x[0], q[1]-x[1], q[2]-x[2] };
    dn = norm3(df); sign = 1.0;
  } else {                                  /* centre inside the ellipsoid: radial projection onto the surface */
    double sc = lev > MINVAL ? 1.0/sqrt(lev) : 0.0;
    if (sc == 0.0) { q[0] = s[0]; q[1] = 0; q[2] = 0; } else for (int k = 0; k < 3; ++k) q[k] = x[k]*sc;
    double df[3] = { q[0]-x[0], q[1]-x[1], q[2]-x[2] };
    dn = norm3(df); sign = -1.0;
  }
  double dd = sign*dn - r;
  if (dd > margin) return 0;
  for (int k = 0; k < 3; ++k) g[k] = -q[k]/(s[k]*s[k]);       /* inward surface normal at q = direction sphere -> ellipsoid */
  double gn = norm3(g), nl[3] = { g[0]/gn, g[1]/gn, g[2]/gn };
  mulmatvec3(nrm, Re, nl);
  *dist = dd;
  for (int k = 0; k < 3; ++k) pos[k] = c[k] + nrm[k]*(r + 0.5*dd);
  return 1;
}
static int collide_pair(const OrcModel* m, const OrcData* d, int g1, int g2, int sub, double margin,
                        double* dist, double* pos, double* nrm /* up to 2 results */) {
  FL(60);              /* geom poses (two 3x3 products, counted here) and the distance test of the pair */
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *p1 = d->geom_xpos + 3*g1, *p2 = d->geom_xpos + 3*g2;
  const double *R1 = d->geom_xmat + 9*g1, *R2 = d->geom_xmat + 9*g2;
  const double *s1 = d->geom_size + 3*g1, *s2 = d->geom_size + 3*g2;
  if (t1 == MYO_GEOM_PLANE && t2 == MYO_GEOM_SPHERE) {
    double n[3] = { R1[2], R1[5], R1[8] };
    double dd = (p2[0]-p1[0])*n[0]+(p2[1]-p1[1])*n[1]+(p2[2]-p1[2])*n[2] - s2[0];
    if (dd > margin) return 0;
    dist[0] = dd; memcpy(nrm, n, sizeof n);
    for (int k = 0; k < 3; ++k) pos[k] = p2[k] - n[k]*(s2[0] + 0.5*dd);
    return 1;
  }
  if (t1 == MYO_GEOM_PLANE && t2 == MYO_GEOM_CAPSULE) {
    double n[3] = { R1[2], R1[5], R1[8] }, ax[3] = { R2[2], R2[5], R2[8] };
    int cnt = 0;
    for (int e = 0; e < 2; ++e) {
      double sg = e ? -1.0 : 1.0, c[3];
      for (int k = 0; k < 3; ++k) c[k] = p2[k] + sg*s2[1]*ax[k];
      double dd = (c[0]-p1[0])*n[0]+(c[1]-p1[1])*n[1]+(c[2]-p1[2])*n[2] - s2[0];
      if (dd > margin) continue;
      dist[cnt] = dd; memcpy(nrm+3*cnt, n, sizeof n);
      for (int k = 0; k < 3; ++k) pos[3*cnt+k] = c[k] - n[k]*(s2[0] + 0.5*dd);
      cnt++;
    }
    return cnt;
  }
  if (t1 == MYO_GEOM_SPHERE && t2 == MYO_GEOM_SPHERE)
    return sphere_sphere_raw(dist, pos, nrm, p1, s1[0], p2, s2[0], margin);
  if (t1 == MYO_GEOM_SPHERE && t2 == MYO_GEOM_CAPSULE) {
    double ax[3] = { R2[2], R2[5], R2[8] }, q[3];
    seg_nearest(q, p2, ax, s2[1], p1);
    return sphere_sphere_raw(dist, pos, nrm, p1, s1[0], q, s2[0], margin);
  }
  if (t1 == MYO_GEOM_CAPSULE && t2 == MYO_GEOM_CAPSULE) {
    double a1[3] = { R1[2], R1[5], R1[8] }, a2[3] = { R2[2], R2[5], R2[8] };
    double dif[3] = { p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2] };
    double ma = dot3(a1,a1), mb = -dot3(a1,a2), mc = dot3(a2,a2);
    double u = -dot3(a1,dif), v = dot3(a2,dif);
    double det = ma*mc - mb*mb, x1, x2;
    if (fabs(det) < 1e-12) { /* parallel: project centre of 2 onto 1 */
      x1 = 0; x2 = v/mc;
    } else {
      x1 = (mc*u - mb*v)/det; x2 = (ma*v - mb*u)/det;
    }
    if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1];
    if (x2 > s2[1]) x2 = s2[1]; else if (x2 < -s2[1]) x2 = -s2[1];
    double q1[3], q2[3];
    for (int k = 0; k < 3; ++k) q1[k] = p1[k] + x1*a1[k];
    seg_nearest(q2, p2, a2, s2[1], q1);
    seg_nearest(q1, p1, a1, s1[1], q2);
    return sphere_sphere_raw(dist, pos, nrm, q1, s1[0], q2, s2[0], margin);
  }
  if (t1 == MYO_GEOM_SPHERE && t2 == MYO_GEOM_BOX) {
    double t[3] = { p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2] }, c[3], cl[3];
    mulmatTvec3(c, R2, t);
    int inside = 1;
    for (int k = 0; k < 3; ++k) {
      cl[k] = c[k];
      if (cl[k] > s2[k]) { cl[k] = s2[k]; inside = 0; } else if (cl[k] < -s2[k]) { cl[k] = -s2[k]; inside = 0; }
    }
    double nl[3], dd;
    if (!inside) {
      double df[3] = { cl[0]-c[0], cl[1]-c[1], cl[2]-c[2] };
      double dn = norm3(df);
      dd = dn - s1[0];
      if (dd > margin) return 0;
      nl[0]=df[0]/dn; nl[1]=df[1]/dn; nl[2]=df[2]/dn;
    } else { /* centre inside the box: push out through the nearest face */
      int kb = 0; double best = 1e300;
      for (int k = 0; k < 3; ++k) { double e = s2[k]-fabs(c[k]); if (e < best) { best = e; kb = k; } }
      nl[0]=nl[1]=nl[2]=0; nl[kb] = c[kb] > 0 ? -1.0 : 1.0;
      dd = -best - s1[0];
    }
    mulmatvec3(nrm, R2, nl);
    dist[0] = dd;
    for (int k = 0; k < 3; ++k) pos[k] = p1[k] + nrm[k]*(s1[0] + 0.5*dd);
    return 1;
  }
  if (t1 == MYO_GEOM_PLANE && t2 == MYO_GEOM_ELLIPSOID) {      /* deepest point of the ellipsoid along -n (support mapping) */
    double n[3] = { R1[2], R1[5], R1[8] }, w[3], sw[3], q[3], ql[3];
    mulmatTvec3(w, R2, n);
    for (int k = 0; k < 3; ++k) sw[k] = s2[k]*w[k];
    double L = norm3(sw);
    for (int k = 0; k < 3; ++k) ql[k] = -s2[k]*sw[k]/L;
    mulmatvec3(q, R2, ql);
    for (int k = 0; k < 3; ++k) q[k] += p2[k];
    double dd = (q[0]-p1[0])*n[0]+(q[1]-p1[1])*n[1]+(q[2]-p1[2])*n[2];
    if (dd > margin) return 0;
    dist[0] = dd; memcpy(nrm, n, sizeof n);
    for (int k = 0; k < 3; ++k) pos[k] = q[k] - n[k]*0.5*dd;
    return 1;
  }
  if (t1 == MYO_GEOM_PLANE && t2 == MYO_GEOM_CYLINDER) {       /* the lowest rim point of each cap (the cap centre when the cap is level) */
    double n[3] = { R1[2], R1[5], R1[8] }, ax[3] = { R2[2], R2[5], R2[8] };
    double na = dot3(n, ax), np_[3] = { n[0]-na*ax[0], n[1]-na*ax[1], n[2]-na*ax[2] };
    double L = norm3(np_);
    int cnt = 0;
    for (int e = 0; e < 2; ++e) {
      double sg = e ? -1.0 : 1.0, q[3];
      for (int k = 0; k < 3; ++k) q[k] = p2[k] + sg*s2[1]*ax[k] - (L > 1e-12 ? s2[0]*np_[k]/L : 0.0);
      double dd = (q[0]-p1[0])*n[0]+(q[1]-p1[1])*n[1]+(q[2]-p1[2])*n[2];
      if (dd > margin) continue;
      dist[cnt] = dd; memcpy(nrm+3*cnt, n, sizeof n);
      for (int k = 0; k < 3; ++k) pos[3*cnt+k] = q[k] - n[k]*0.5*dd;
      cnt++;
    }
    return cnt;
  }
  if (t1 == MYO_GEOM_SPHERE && t2 == MYO_GEOM_CYLINDER) return point_cylinder(dist, pos, nrm, p1, s1[0], p2, R2, s2[0], s2[1], margin);
  if (t1 == MYO_GEOM_SPHERE && t2 == MYO_GEOM_ELLIPSOID) return point_ellipsoid(dist, pos, nrm, p1, s1[0], p2, R2, s2, margin);
  if (t1 == MYO_GEOM_CAPSULE && (t2 == MYO_GEOM_CYLINDER || t2 == MYO_GEOM_BOX || t2 == MYO_GEOM_ELLIPSOID)) {
    /* the point of the capsule's segment nearest to geom 2, then a sphere of the capsule's radius there */
    double ax[3] = { R1[2], R1[5], R1[8] }, t[3] = { p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2] }, c[3], a[3], ts;
    mulmatTvec3(c, R2, t); mulmatTvec3(a, R2, ax);
    if (t2 == MYO_GEOM_ELLIPSOID) {       /* nearest in the metric that makes the ellipsoid a unit sphere: closed form */
      double num = 0, den = 0;
      for (int k = 0; k < 3; ++k) { num += c[k]*a[k]/(s2[k]*s2[k]); den += a[k]*a[k]/(s2[k]*s2[k]); }
      ts = den > MINVAL ? -num/den : 0.0;
      if (ts > s1[1]) ts = s1[1]; else if (ts < -s1[1]) ts = -s1[1];
    } else if (t2 == MYO_GEOM_BOX) {
      SEG_ARGMIN(ts, s1[1], dsd_box(c, a, tt, s2))
    } else {
      SEG_ARGMIN(ts, s1[1], dsd_cylinder(c, a, tt, s2[0], s2[1]))
    }
    double q[3] = { p1[0]+ts*ax[0], p1[1]+ts*ax[1], p1[2]+ts*ax[2] };
    if (t2 == MYO_GEOM_CYLINDER) return point_cylinder(dist, pos, nrm, q, s1[0], p2, R2, s2[0], s2[1], margin);
    if (t2 == MYO_GEOM_ELLIPSOID) return point_ellipsoid(dist, pos, nrm, q, s1[0], p2, R2, s2, margin);
    /* capsule-box: a capsule lying on a face rests on its two ends (both within the margin: two contacts, like plane-capsule);
     * otherwise one contact at the nearest point of the segment */
    double qa[3] = { p1[0]+s1[1]*ax[0], p1[1]+s1[1]*ax[1], p1[2]+s1[1]*ax[2] }, qb[3] = { p1[0]-s1[1]*ax[0], p1[1]-s1[1]*ax[1], p1[2]-s1[1]*ax[2] };
    int cnt = point_box(dist, pos, nrm, qa, s1[0], p2, R2, s2, margin, 1);
    if (cnt) cnt += point_box(dist+1, pos+3, nrm+3, qb, s1[0], p2, R2, s2, margin, 1);
    if (cnt == 2) return 2;
    return point_box(dist, pos, nrm, q, s1[0], p2, R2, s2, margin, 1);
  }
  if (t1 == MYO_GEOM_BOX && t2 == MYO_GEOM_BOX && sub == 17) {
    /* edge-edge contact (the 17th candidate of a box-box pair): separating-axis test over the 6 face normals and the 9 edge
     * cross products; when the axis of LARGEST separation (least penetration) is an edge pair, one contact at the midpoint
     * of the two edges' closest points, normal = that axis, from box 1 to box 2.  Face contacts are the vertex-face
     * candidates' business: a face axis wins ties (the 1e-9 relative bias keeps parallel-edge configurations out). */
    double t[3] = { p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2] };
    double A[3][3], B[3][3];
    for (int k = 0; k < 3; ++k) for (int e = 0; e < 3; ++e) { A[k][e] = R1[3*e+k]; B[k][e] = R2[3*e+k]; }   /* box axes = matrix columns */
    double best_face = -1e300, best_edge = -1e300, Ln[3] = {0,0,0}; int bi = -1, bj = -1;
    for (int k = 0; k < 6; ++k) {
      const double* L = k < 3 ? A[k] : B[k-3];
      double ra = 0, rb = 0;
      for (int e = 0; e < 3; ++e) { ra += s1[e]*fabs(dot3(A[e], L)); rb += s2[e]*fabs(dot3(B[e], L)); }
      double sep = fabs(dot3(t, L)) - ra - rb;
      if (sep > best_face) best_face = sep;
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double L[3]; cross3(L, A[i], B[j]);
      double len = sqrt(dot3(L, L));
      if (len < 1e-6) continue;                                   /* (nearly) parallel edges: a face axis covers them */
      for (int e = 0; e < 3; ++e) L[e] /= len;
      double ra = 0, rb = 0;
      for (int e = 0; e < 3; ++e) { ra += s1[e]*fabs(dot3(A[e], L)); rb += s2[e]*fabs(dot3(B[e], L)); }
      double sep = fabs(dot3(t, L)) - ra - rb;
      if (sep > best_edge) { best_edge = sep; bi = i; bj = j; for (int e = 0; e < 3; ++e) Ln[e] = L[e]; }
    }
    if (bi < 0 || best_edge > margin || best_face > margin) return 0;
    if (!(best_edge > best_face + 1e-9*(1.0 + fabs(best_face)))) return 0;
    if (dot3(t, Ln) < 0) for (int e = 0; e < 3; ++e) Ln[e] = -Ln[e];                    /* from box 1 to box 2 */
    /* the edge of box 1 parallel to A[bi] that is extreme along +Ln, the edge of box 2 parallel to B[bj] extreme along -Ln */
    double ea[3] = { p1[0], p1[1], p1[2] }, eb[3] = { p2[0], p2[1], p2[2] };
    for (int k = 0; k < 3; ++k) {
      if (k != bi) { double sg = dot3(A[k], Ln) > 0 ? 1.0 : -1.0; for (int e = 0; e < 3; ++e) ea[e] += sg*s1[k]*A[k][e]; }
      if (k != bj) { double sg = dot3(B[k], Ln) > 0 ? -1.0 : 1.0; for (int e = 0; e < 3; ++e) eb[e] += sg*s2[k]*B[k][e]; }
    }
    /* closest points of the lines ea + u A[bi], eb + v B[bj], clamped to the edges */
    double w[3] = { ea[0]-eb[0], ea[1]-eb[1], ea[2]-eb[2] };
    double ab = dot3(A[bi], B[bj]), aw = dot3(A[bi], w), bw = dot3(B[bj], w), den = 1.0 - ab*ab;
    double u = den > 1e-12 ? (ab*bw - aw)/den : 0.0, v = den > 1e-12 ? (bw - ab*aw)/den : 0.0;
    u = u < -s1[bi] ? -s1[bi] : (u > s1[bi] ? s1[bi] : u);
    v = v < -s2[bj] ? -s2[bj] : (v > s2[bj] ? s2[bj] : v);
    for (int e = 0; e < 3; ++e) { pos[e] = 0.5*((ea[e] + u*A[bi][e]) + (eb[e] + v*B[bj][e])); nrm[e] = Ln[e]; }
    dist[0] = best_edge;
    return 1;
  }
  if (t1 == MYO_GEOM_BOX && t2 == MYO_GEOM_BOX && sub >= 1 && sub <= 16) {
    /* vertex-face contacts: the model compiler lists a box-box geom pair as 16 candidates, sub = 1 + v: vertex v (sign bits) of
     * box 1 against box 2, sub = 9 + v: vertex v of box 2 against box 1; sub = 17: the edge-edge candidate above. */
    int v = (sub - 1) & 7, second = sub > 8;
    const double *pv = second ? p2 : p1, *Rv = second ? R2 : R1, *sv = second ? s2 : s1;
    const double *pb = second ? p1 : p2, *Rb = second ? R1 : R2, *sb = second ? s1 : s2;
    double loc[3] = { (v & 1) ? sv[0] : -sv[0], (v & 2) ? sv[1] : -sv[1], (v & 4) ? sv[2] : -sv[2] }, q[3];
    mulmatvec3(q, Rv, loc);
    for (int k = 0; k < 3; ++k) q[k] += pv[k];
    int cnt = point_box(dist, pos, nrm, q, 0.0, pb, Rb, sb, margin, 0);
    if (cnt && second) for (int k = 0; k < 3; ++k) nrm[k] = -nrm[k];      /* the normal runs from geom 1 to geom 2 */
    return cnt;
  }
  return 0;
}

static void mix_params(const OrcModel* m, const OrcData* d, int g1, int g2, OrcContact* c) {
  int pr1 = m->geom_priority[g1], pr2 = m->geom_priority[g2];
  double mix;
  if (pr1 != pr2) mix = pr1 > pr2 ? 1.0 : 0.0;
  else {
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1/(s1+s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
  }
  const double *r1 = m->geom_solref + 2*g1, *r2 = m->geom_solref + 2*g2;
  if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; ++k) c->solref[k] = mix*r1[k] + (1-mix)*r2[k];
  else for (int k = 0; k < 2; ++k) c->solref[k] = r1[k] < r2[k] ? r1[k] : r2[k];
  for (int k = 0; k < 5; ++k) c->solimp[k] = mix*m->geom_solimp[5*g1+k] + (1-mix)*m->geom_solimp[5*g2+k];
  double f[3];
  for (int k = 0; k < 3; ++k) {
    double a = d->geom_friction[3*g1+k], b = d->geom_friction[3*g2+k];
    f[k] = (pr1 == pr2) ? (a > b ? a : b) : (pr1 > pr2 ? a : b);
  }
  c->friction[0]=f[0]; c->friction[1]=f[0]; c->friction[2]=f[1]; c->friction[3]=f[2]; c->friction[4]=f[2];
  /* mj_contactParam: condim of the geom with the higher priority, the larger of the two at equal priority */
  int d1 = m->geom_condim[g1], d2 = m->geom_condim[g2];
  c->dim = (pr1 == pr2) ? (d1 > d2 ? d1 : d2) : (pr1 > pr2 ? d1 : d2);
}

static void collision(const OrcModel* m, OrcData* d) {
  d->ncon = 0;
  for (int p = 0; p < m->npair; ++p) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    double mg1 = m->geom_margin[g1], mg2 = m->geom_margin[g2];
    double margin = mg1 > mg2 ? mg1 : mg2;
    double gap = m->geom_gap[g1] > m->geom_gap[g2] ? m->geom_gap[g1] : m->geom_gap[g2];
    int xp = m->pair_explicit ? m->pair_explicit[p] : -1;
    if (xp >= 0) { margin = m->xp_margin[xp]; gap = m->xp_gap[xp]; }      /* an explicit <pair>: its own margin and gap */
    /* bounding-sphere filter with the MODEL's rbound: the reference rewrites geom_size per
     * episode without refreshing rbound (baoding.py:586-604), so the stale value gates contacts */
    double rb1 = m->geom_rbound[g1], rb2 = m->geom_rbound[g2];
    if (rb1 > 0 && rb2 > 0) {
      double df[3] = { d->geom_xpos[3*g1]-d->geom_xpos[3*g2], d->geom_xpos[3*g1+1]-d->geom_xpos[3*g2+1],
                       d->geom_xpos[3*g1+2]-d->geom_xpos[3*g2+2] };
      double bound = rb1 + rb2 + margin;
      if (dot3(df, df) > bound*bound) continue;
    }
    double dist[2], pos[6], nrm[6];
    int n = collide_pair(m, d, g1, g2, m->pair_sub ? m->pair_sub[p] : 0, margin, dist, pos, nrm);
    for (int k = 0; k < n && d->ncon < MAXCON; ++k) {
      OrcContact* c = &d->con[d->ncon++];
      c->dist = dist[k]; memcpy(c->pos, pos+3*k, 3*sizeof(double));
      memcpy(c->frame, nrm+3*k, 3*sizeof(double)); make_frame(c->frame);
      c->includemargin = margin - gap; c->geom1 = g1; c->geom2 = g2;
      mix_params(m, d, g1, g2, c);
      if (xp >= 0) {                       /* mj_contactParam for a predefined pair: the pair's parameters, nothing mixed */
        memcpy(c->solref, m->xp_solref + 2*xp, 2*sizeof(double)); memcpy(c->solimp, m->xp_solimp + 5*xp, 5*sizeof(double));
        const double* f = m->xp_friction + 3*xp;
        c->friction[0] = f[0]; c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = f[2]; c->friction[4] = f[2];
        c->dim = m->xp_dim[xp];
      }
    }
  }
}

/* rotational Jacobian of a body: column i = the angular part of dof i's motion axis for the body's ancestor dofs */
static void jac_rot(const OrcModel* m, const OrcData* d, int body, double* jacr /* 3 x nv */) {
  int nv = m->nv;
  memset(jacr, 0, sizeof(double)*3*nv);
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    const double* cd = d->cdof + 6*i;
    jacr[i] = cd[0]; jacr[nv+i] = cd[1]; jacr[2*nv+i] = cd[2];
    i = m->dof_parentid[i];
  }
}

/* ------------------------------------------------------------------ P7 constraints */
static void get_solparam(const OrcModel* m, const double* solref, const double* solimp, double pos_minus_margin,
                         double* K, double* B, double* I) {
  double d0 = solimp[0], d1 = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (d0 < MINIMP) d0 = MINIMP; else if (d0 > MAXIMP) d0 = MAXIMP;
  if (d1 < MINIMP) d1 = MINIMP; else if (d1 > MAXIMP) d1 = MAXIMP;
  if (width < 0) width = 0;
  if (mid < MINIMP) mid = MINIMP; else if (mid > MAXIMP) mid = MAXIMP;
  if (power < 1) power = 1;
  double imp;
  if (d0 == d1 || width <= MINVAL) imp = 0.5*(d0+d1);
  else {
    double x = fabs(pos_minus_margin)/width;
    if (x >= 1) imp = d1;
    else if (x <= 0) imp = d0;
    else {
      double y;
      if (power == 1) y = x;
      else if (x <= mid) y = pow(x, power)/pow(mid, power-1);
      else y = 1 - pow(1-x, power)/pow(1-mid, power-1);
      imp = d0 + y*(d1-d0);
    }
  }
  double tc = solref[0], dr = solref[1];
  if (tc > 0) {
    if (!(m->disableflags & (1<<11)) && tc < 2*m->timestep) tc = 2*m->timestep; /* refsafe */
    double a = d1*d1*tc*tc*dr*dr; *K = 1/(a > MINVAL ? a : MINVAL);
    double b = d1*tc; *B = 2/(b > MINVAL ? b : MINVAL);
  } else { *K = -tc/(d1*d1); *B = -dr/d1; }
  *I = imp;
}

static void add_row(OrcData* d, int nv, int type, int id, double pos, double margin, double diagApprox,
                    const double* solref, const double* solimp) {
  int r = d->nefc;
  if (r >= MAXEFC) return;
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  d->efc_diagApprox[r] = diagApprox; d->efc_floss[r] = 0;
  double K, B, I;
  get_solparam(d->m, solref, solimp, pos - margin, &K, &B, &I);
  d->efc_KBIP[4*r] = K; d->efc_KBIP[4*r+1] = B; d->efc_KBIP[4*r+2] = I; d->efc_KBIP[4*r+3] = 0;
  double R = (1-I)*diagApprox/I; if (R < MINVAL) R = MINVAL;
  d->efc_R[r] = R;
  d->nefc++;
  (void)nv;
}

static void make_constraint(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  d->nefc = 0; d->nl = 0; d->ntl = 0;
  double* jac1 = d->w1; double* jac2 = d->w2;
  /* friction loss (mj_instantiateFriction): one row per dof / tendon with frictionloss > 0, J = the dof / the tendon's moment arms,
   * pos = margin = 0; the row's force is clamped to +- frictionloss (constraint_update), its cost is Huber-shaped */
  int nf = 0;
  if (m->dof_frictionloss) for (int i = 0; i < nv; ++i) {
    if (!(m->dof_frictionloss[i] > 0) || d->nefc >= MAXEFC) continue;
    double* J = d->efc_J + d->nefc*nv; memset(J, 0, sizeof(double)*nv);
    J[i] = 1;
    add_row(d, nv, 3, i, 0.0, 0.0, m->dof_invweight0[i], m->dof_solref + 2*i, m->dof_solimp + 5*i);
    d->efc_floss[d->nefc-1] = m->dof_frictionloss[i]; nf++;
  }
  if (m->tendon_frictionloss) for (int t = 0; t < m->ntendon; ++t) {
    if (!(m->tendon_frictionloss[t] > 0) || d->nefc >= MAXEFC) continue;
    double* J = d->efc_J + d->nefc*nv;
    for (int c = 0; c < nv; ++c) J[c] = d->ten_J[t*nv+c];
    add_row(d, nv, 3, t, 0.0, 0.0, m->tendon_invweight0[t], m->tendon_solref_fri + 2*t, m->tendon_solimp_fri + 5*t);
    d->efc_floss[d->nefc-1] = m->tendon_frictionloss[t]; nf++;
  }
  (void)nf;
  /* joint limits */
  for (int j = 0; j < m->njnt; ++j) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == MYO_JNT_FREE) continue;
    double q = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? q - m->jnt_range[2*j] : m->jnt_range[2*j+1] - q;
      if (dist < m->jnt_margin[j] && d->nefc < MAXEFC) {
        double* J = d->efc_J + d->nefc*nv; memset(J, 0, sizeof(double)*nv);
        J[m->jnt_dofadr[j]] = -(double)side;
        add_row(d, nv, 0, j, dist, m->jnt_margin[j], m->dof_invweight0[m->jnt_dofadr[j]],
                m->jnt_solref + 2*j, m->jnt_solimp + 5*j);
        d->nl++;
      }
    }
  }
  /* tendon limits */
  for (int t = 0; t < m->ntendon; ++t) {
    if (!m->tendon_limited[t]) continue;
    double L = d->ten_length[t];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? L - m->tendon_range[2*t] : m->tendon_range[2*t+1] - L;
      if (dist < m->tendon_margin[t] && d->nefc < MAXEFC) {
        double* J = d->efc_J + d->nefc*nv;
        for (int c = 0; c < nv; ++c) J[c] = -(double)side*d->ten_J[t*nv+c];
        add_row(d, nv, 1, t, dist, m->tendon_margin[t], m->tendon_invweight0[t],
                m->tendon_solref_lim + 2*t, m->tendon_solimp_lim + 5*t);
        d->ntl++;
      }
    }
  }
  /* contacts: condim 1 -> one row; condim 3 / 4 / 6 -> pyramidal, 4 / 6 / 10 rows */
  for (int ci = 0; ci < d->ncon; ++ci) {
    OrcContact* c = &d->con[ci];
    if (c->dist >= c->includemargin) continue;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    jac_point(m, d, b1, c->pos, jac1); jac_point(m, d, b2, c->pos, jac2);
    double Jf[3][64*4]; /* nv <= 256 */
    { int nz = 0; for (int col = 0; col < nv; ++col) if (jac2[col] != jac1[col] || jac2[nv+col] != jac1[nv+col] || jac2[2*nv+col] != jac1[2*nv+col]) nz++;
      FL(nz*(3 + 3*5) + 4*nz*2 + 12); }      /* frame-rotated relative Jacobian and the four pyramid rows, structurally non-zero columns */
    for (int a = 0; a < 3; ++a)
      for (int col = 0; col < nv; ++col)
        Jf[a][col] = c->frame[3*a]*(jac2[col]-jac1[col]) + c->frame[3*a+1]*(jac2[nv+col]-jac1[nv+col]) +
                     c->frame[3*a+2]*(jac2[2*nv+col]-jac1[2*nv+col]);
    double tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
    double rot = m->body_invweight0[2*b1+1] + m->body_invweight0[2*b2+1];
    int first = d->nefc, dim = c->dim;
    if (dim == 1) {                       /* frictionless: the normal row alone, no pyramid */
      if (d->nefc < MAXEFC) {
        double* J = d->efc_J + d->nefc*nv;
        for (int col = 0; col < nv; ++col) J[col] = Jf[0][col];
        add_row(d, nv, 2, ci, c->dist, c->includemargin, tran, c->solref, c->solimp);
      }
      continue;
    }
    /* pyramidal cone of dimension dim: two rows  J_normal +- friction[k] J_k  per direction k < dim - 1; directions 0, 1 are the two
     * tangential translations, 2 the rotation about the normal (torsional friction), 3, 4 the rotations about the tangents (rolling) */
    double Jr[3][64*4];
    if (dim > 3) {
      double* jr1 = d->w3; double* jr2 = d->w4;
      jac_rot(m, d, b1, jr1); jac_rot(m, d, b2, jr2);
      for (int a = 0; a < 3; ++a)
        for (int col = 0; col < nv; ++col)
          Jr[a][col] = c->frame[3*a]*(jr2[col]-jr1[col]) + c->frame[3*a+1]*(jr2[nv+col]-jr1[nv+col]) + c->frame[3*a+2]*(jr2[2*nv+col]-jr1[2*nv+col]);
    }
    for (int k = 0; k < dim - 1; ++k) {
      double mu = c->friction[k];
      const double* Jk = k < 2 ? Jf[1+k] : Jr[k-2];
      for (int sg = 0; sg < 2; ++sg) {
        if (d->nefc >= MAXEFC) break;
        double* J = d->efc_J + d->nefc*nv;
        double s = sg ? -mu : mu;
        for (int col = 0; col < nv; ++col) J[col] = Jf[0][col] + s*Jk[col];
        add_row(d, nv, 2, ci, c->dist, c->includemargin, tran + mu*mu*(k < 2 ? tran : rot), c->solref, c->solimp);
      }
    }
    /* pyramidal regularisation: every edge row gets Rpy = 2 mu^2 R(first row), mu = friction[0]/sqrt(impratio) */
    if (d->nefc - first == 2*(dim - 1)) {
      double mu = c->friction[0]/sqrt(m->impratio);
      double Rpy = 2*mu*mu*d->efc_R[first];
      if (Rpy < MINVAL) Rpy = MINVAL;
      for (int r = first; r < d->nefc; ++r) d->efc_R[r] = Rpy;
    }
  }
  for (int r = 0; r < d->nefc; ++r) d->efc_D[r] = 1/d->efc_R[r];
}

/* ------------------------------------------------------------------ P8 velocity stage */
static void fwd_velocity(const OrcModel* m, OrcData* d) {
  int nv = m->nv, nb = m->nbody;
  for (int t = 0; t < m->ntendon; ++t) {
    double s = 0; for (int c = 0; c < nv; ++c) { s += d->ten_J[t*nv+c]*d->qvel[c]; if (d->ten_J[t*nv+c] != 0) FL(2); }
    d->ten_velocity[t] = s;
  }
  for (int i = 0; i < m->nu; ++i) {
    double s = 0; for (int c = 0; c < nv; ++c) { s += d->actuator_moment[i*nv+c]*d->qvel[c]; if (d->actuator_moment[i*nv+c] != 0) FL(2); }
    d->actuator_velocity[i] = s;
  }
  /* comVel */
  memset(d->cvel, 0, 6*sizeof(double));
  for (int b = 1; b < nb; ++b) {
    double cv[6]; memcpy(cv, d->cvel + 6*m->body_parentid[b], sizeof cv);
    int jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    for (int k = 0; k < jn; ++k) {
      int j = ja + k, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == MYO_JNT_FREE) {
        memset(d->cdof_dot + 6*da, 0, 18*sizeof(double));
        for (int a = 0; a < 3; ++a) for (int e = 0; e < 6; ++e) cv[e] += d->cdof[6*(da+a)+e]*d->qvel[da+a];
        for (int a = 3; a < 6; ++a) cross_motion(d->cdof_dot + 6*(da+a), cv, d->cdof + 6*(da+a));
        for (int a = 3; a < 6; ++a) for (int e = 0; e < 6; ++e) cv[e] += d->cdof[6*(da+a)+e]*d->qvel[da+a];
      } else {
        cross_motion(d->cdof_dot + 6*da, cv, d->cdof + 6*da);
        for (int e = 0; e < 6; ++e) cv[e] += d->cdof[6*da+e]*d->qvel[da];
      }
    }
    FL(12 * m->body_dofnum[b]);
    memcpy(d->cvel + 6*b, cv, sizeof cv);
  }
  /* passive */
  FL(nv + 3*m->njnt);
  for (int c = 0; c < nv; ++c) d->qfrc_passive[c] = -m->dof_damping[c]*d->qvel[c];
  for (int j = 0; j < m->njnt; ++j) {
    if (m->jnt_type[j] == MYO_JNT_FREE || m->jnt_stiffness[j] == 0) continue;
    int qa = m->jnt_qposadr[j];
    d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j]*(d->qpos[qa] - m->qpos_spring[qa]);
  }
  for (int t = 0; t < m->ntendon; ++t) {
    double k = m->tendon_stiffness[t], b = m->tendon_damping[t];
    if (k == 0 && b == 0) continue;
    double f = -k*(d->ten_length[t] - m->tendon_lengthspring[t]) - b*d->ten_velocity[t];
    for (int c = 0; c < nv; ++c) d->qfrc_passive[c] += d->ten_J[t*nv+c]*f;
  }
  /* constraint reference needs efc_vel */
  for (int r = 0; r < d->nefc; ++r) {
    double s = 0; for (int c = 0; c < nv; ++c) { s += d->efc_J[r*nv+c]*d->qvel[c]; if (d->efc_J[r*nv+c] != 0) FL(2); }
    FL(6);
    d->efc_vel[r] = s;
    d->efc_aref[r] = -d->efc_KBIP[4*r+1]*s - d->efc_KBIP[4*r]*d->efc_KBIP[4*r+2]*(d->efc_pos[r] - d->efc_margin[r]);
  }
  /* RNE bias */
  double* cacc = d->cacc; double* cfrc = d->cfrc;
  cacc[0]=cacc[1]=cacc[2]=0; cacc[3]=-m->gravity[0]; cacc[4]=-m->gravity[1]; cacc[5]=-m->gravity[2];
  memset(cfrc, 0, 6*sizeof(double));
  for (int b = 1; b < nb; ++b) {
    double a[6]; memcpy(a, cacc + 6*m->body_parentid[b], sizeof a);
    int da = m->body_dofadr[b];
    for (int k = 0; k < m->body_dofnum[b]; ++k)
      for (int e = 0; e < 6; ++e) a[e] += d->cdof_dot[6*(da+k)+e]*d->qvel[da+k];
    memcpy(cacc + 6*b, a, sizeof a);
    double t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert + 10*b, a);
    mul_inert_vec(t2, d->cinert + 10*b, d->cvel + 6*b);
    cross_force(t3, d->cvel + 6*b, t2);
    for (int e = 0; e < 6; ++e) cfrc[6*b+e] = t1[e] + t3[e];
    FL(12 * m->body_dofnum[b] + 6 + 6);
  }
  for (int b = nb-1; b > 0; --b) {
    int p = m->body_parentid[b];
    if (p > 0) for (int e = 0; e < 6; ++e) cfrc[6*p+e] += cfrc[6*b+e];
  }
  for (int c = 0; c < nv; ++c) {
    double s = 0; for (int e = 0; e < 6; ++e) s += d->cdof[6*c+e]*cfrc[6*m->dof_bodyid[c]+e];
    FL(12);
    d->qfrc_bias[c] = s;
  }
}

/* ------------------------------------------------------------------ P9 actuation */
static double muscle_gain_length(double L, double lmin, double lmax) {
  if (L < lmin || L > lmax) return 0;
  double a = 0.5*(lmin+1), b = 0.5*(1+lmax), x;
  if (L <= a) { x = (L-lmin)/fmax(MINVAL, a-lmin); return 0.5*x*x; }
  if (L <= 1) { x = (1-L)/fmax(MINVAL, 1-a); return 1-0.5*x*x; }
  if (L <= b) { x = (L-1)/fmax(MINVAL, b-1); return 1-0.5*x*x; }
  x = (lmax-L)/fmax(MINVAL, lmax-b); return 0.5*x*x;
}
static double muscle_gain(double len, double vel, const double* lr, double acc0, const double* prm) {
  double range0 = prm[0], range1 = prm[1], force = prm[2], scale = prm[3], lmin = prm[4], lmax = prm[5],
         vmax = prm[6], fvmax = prm[8];
  if (force < 0) force = scale/fmax(MINVAL, acc0);
  double L0 = (lr[1]-lr[0])/fmax(MINVAL, range1-range0);
  double L = range0 + (len-lr[0])/fmax(MINVAL, L0);
  double V = vel/fmax(MINVAL, L0*vmax);
  double FL = muscle_gain_length(L, lmin, lmax), FV;
  double y = fvmax-1;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V+1)*(V+1);
  else if (V <= y) FV = fvmax - (y-V)*(y-V)/fmax(MINVAL, y);
  else FV = fvmax;
  return -force*FL*FV;
}
static double muscle_bias(double len, const double* lr, double acc0, const double* prm) {
  double range0 = prm[0], range1 = prm[1], force = prm[2], scale = prm[3], lmax = prm[5], fpmax = prm[7];
  if (force < 0) force = scale/fmax(MINVAL, acc0);
  double L0 = (lr[1]-lr[0])/fmax(MINVAL, range1-range0);
  double L = range0 + (len-lr[0])/fmax(MINVAL, L0);
  double b = 0.5*(1+lmax);
  if (L <= 1) return 0;
  if (L <= b) { double x = (L-1)/fmax(MINVAL, b-1); return -force*fpmax*0.5*x*x; }
  double x = (L-b)/fmax(MINVAL, b-1);
  return -force*fpmax*(0.5+x);
}

static void fwd_actuation(const OrcModel* m, OrcData* d) {
  FL(70 * m->nu);      /* activation dynamics, length / velocity normalisation, FL, FV, passive force per muscle */
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(double)*nv);
  for (int i = 0; i < m->nu; ++i) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i]) {
      double lo = m->actuator_ctrlrange[2*i], hi = m->actuator_ctrlrange[2*i+1];
      if (ctrl < lo) ctrl = lo; else if (ctrl > hi) ctrl = hi;
    }
    double input = ctrl;
    if (m->actuator_dyntype[i] == MYO_DYN_MUSCLE) {
      int ia = i - (m->nu - m->na);
      double act = d->act[ia];
      const double* prm = m->actuator_dynprm + 10*i;
      double cc = ctrl < 0 ? 0 : (ctrl > 1 ? 1 : ctrl), ac = act < 0 ? 0 : (act > 1 ? 1 : act);
      double tau = cc > act ? prm[0]*(0.5+1.5*ac) : prm[1]/(0.5+1.5*ac);
      d->act_dot[ia] = (cc-act)/fmax(MINVAL, tau);
      input = act;
    }
    double len = d->actuator_length[i], vel = d->actuator_velocity[i], gain, bias = 0;
    if (m->actuator_gaintype[i] == MYO_GAIN_MUSCLE)
      gain = muscle_gain(len, vel, m->actuator_lengthrange + 2*i, m->actuator_acc0[i], m->actuator_gainprm + 10*i);
    else gain = m->actuator_gainprm[10*i];
    if (m->actuator_biastype[i] == MYO_BIAS_MUSCLE)
      bias = muscle_bias(len, m->actuator_lengthrange + 2*i, m->actuator_acc0[i], m->actuator_biasprm + 10*i);
    else if (m->actuator_biastype[i] == MYO_BIAS_AFFINE)
      bias = m->actuator_biasprm[10*i] + m->actuator_biasprm[10*i+1]*len + m->actuator_biasprm[10*i+2]*vel;
    double f = gain*input + bias;
    if (m->actuator_forcelimited[i]) {
      double lo = m->actuator_forcerange[2*i], hi = m->actuator_forcerange[2*i+1];
      if (f < lo) f = lo; else if (f > hi) f = hi;
    }
    d->actuator_force[i] = f;
    for (int c = 0; c < nv; ++c) d->qfrc_actuator[c] += d->actuator_moment[i*nv+c]*f;
  }
}

/* ------------------------------------------------------------------ P10 acceleration + Newton */
static void mul_M(const OrcData* d, int nv, double* r, const double* v) {
  for (int i = 0; i < nv; ++i) { double s = 0; for (int j = 0; j < nv; ++j) { s += d->M[i*nv+j]*v[j]; if (d->M[i*nv+j] != 0) FL(2); } r[i] = s; }
}

/* cost, force and curvature of one row at jar = x.  Ordinary rows (limits, contacts): 0.5 D x^2 where x < 0.  Friction-loss rows
 * (f > 0): quadratic inside |x| < R f, linear outside — the force -D x saturates at +- f (mj_constraintUpdate) */
static double row_cost(double D, double f, double x, double* force, int* quad) {
  if (f > 0) {
    double Rf = f/D;
    if (x <= -Rf) { *force = f; *quad = 0; return f*(-0.5*Rf - x); }
    if (x >= Rf) { *force = -f; *quad = 0; return f*(-0.5*Rf + x); }
    *force = -D*x; *quad = 1; return 0.5*D*x*x;
  }
  if (x < 0) { *force = -D*x; *quad = 1; return 0.5*D*x*x; }
  *force = 0; *quad = 0; return 0;
}

typedef struct { double cost, d1, d2; } LsEval;

static LsEval ls_eval(const OrcData* d, double alpha, const double* jar, const double* jv, const double* qg) {
  LsEval e; e.cost = alpha*alpha*qg[2] + alpha*qg[1] + qg[0]; e.d1 = 2*alpha*qg[2] + qg[1]; e.d2 = 2*qg[2];
  for (int r = 0; r < d->nefc; ++r) {
    double x = jar[r] + alpha*jv[r];
    FL(2);
    if (d->efc_floss[r] > 0) {
      double force; int quad;
      e.cost += row_cost(d->efc_D[r], d->efc_floss[r], x, &force, &quad);
      e.d1 += -force*jv[r];
      if (quad) e.d2 += d->efc_D[r]*jv[r]*jv[r];
    } else if (x < 0) { double D = d->efc_D[r]; e.cost += 0.5*D*x*x; e.d1 += D*x*jv[r]; e.d2 += D*jv[r]*jv[r]; FL(10); }
  }
  return e;
}

/* ---- the line search of mj_solNewton as MuJoCo 2.1 structures it (engine_solver.c: PrimalSearch) [3P-RECALL], behind a switch
   (orc_set_line_search(1) / MYO_ORACLE_LS=primal): the default search above is a safeguarded Newton iteration on p'(alpha) that stops
   inside the same gradient tolerance, so the two return step lengths that differ by O(gtol / p'') — VERDICT r04 item 10 asks how far
   that moves qacc (tools/oracle_linesearch.py, tests/test_oracle_closed_forms.py).  Structure restated here:
     p0 = eval(0); p1 = eval(Newton step from p0); keep the cheaper of the two as p1; converged if |p1'| < gtol
     dir = sign of the descent from p1; one-sided Newton steps from p1 while the derivative keeps its sign (p2 = the previous point):
       converged at any of them -> return it
     the derivative changed sign: bracket [p2, p1]; then per iteration THREE candidates — the Newton points of both bracket ends and
       the midpoint — a converged candidate (|p'| < gtol) of lowest cost is returned, otherwise each end moves to the candidate that
       tightens it most; no end moved, or the iteration budget is spent -> the end of lower cost. */
static int g_ls_mode = -1;
void orc_set_line_search(int primal) { g_ls_mode = primal ? 1 : 0; }
static int ls_mode(void) {
  if (g_ls_mode < 0) { const char* e = getenv("MYO_ORACLE_LS"); g_ls_mode = (e && e[0] == 'p') ? 1 : 0; }
  return g_ls_mode;
}
typedef struct { double alpha; LsEval e; } LsPoint;
static LsPoint ls_point(const OrcData* d, double alpha, const double* jar, const double* jv, const double* qg, int* evals) {
  LsPoint p; p.alpha = alpha; p.e = ls_eval(d, alpha, jar, jv, qg); (*evals)++; return p;
}
static double primal_search(const OrcData* d, const double* jar, const double* jv, const double* qg, double gtol) {
  int it = 0;
  LsPoint p0 = ls_point(d, 0, jar, jv, qg, &it);
  LsPoint p1 = ls_point(d, p0.alpha - p0.e.d1/p0.e.d2, jar, jv, qg, &it);
  if (p0.e.cost < p1.e.cost) p1 = p0;
  if (fabs(p1.e.d1) < gtol) return p1.alpha;
  const double dir = p1.e.d1 < 0 ? 1 : -1;
  LsPoint p2 = p1; int p2update = 0;
  while (p1.e.d1*dir <= -gtol && it < LS_ITER) {
    p2 = p1; p2update = 1;
    p1 = ls_point(d, p1.alpha - p1.e.d1/p1.e.d2, jar, jv, qg, &it);
    if (fabs(p1.e.d1) < gtol) return p1.alpha;
  }
  if (it >= LS_ITER || !p2update) return p1.alpha;
  /* bracket: p2 on the descending side (p2' dir < 0), p1 beyond the minimum */
  LsPoint p1next = ls_point(d, p1.alpha - p1.e.d1/p1.e.d2, jar, jv, qg, &it);
  LsPoint p2next = p1;        /* (p1 is the Newton point of p2) */
  while (it < LS_ITER) {
    LsPoint pmid = ls_point(d, 0.5*(p1.alpha + p2.alpha), jar, jv, qg, &it);
    LsPoint cand[3]; cand[0] = p1next; cand[1] = p2next; cand[2] = pmid;
    int best = -1;
    for (int k = 0; k < 3; ++k) if (fabs(cand[k].e.d1) < gtol && (best < 0 || cand[k].e.cost < cand[best].e.cost)) best = k;
    if (best >= 0) return cand[best].alpha;
    int b1 = 0, b2 = 0;
    for (int k = 0; k < 3; ++k) {
      const double lo = p1.alpha < p2.alpha ? p1.alpha : p2.alpha, hi = p1.alpha < p2.alpha ? p2.alpha : p1.alpha;
      if (!(cand[k].alpha > lo && cand[k].alpha < hi)) continue;                 /* only points inside the bracket tighten it */
      if (cand[k].e.d1*dir > 0) { p1 = cand[k]; b1 = 1; }                        /* beyond the minimum: the far end */
      else { p2 = cand[k]; b2 = 2; }                                             /* still descending: the near end */
    }
    if (!b1 && !b2) break;
    if (b1) p1next = ls_point(d, p1.alpha - p1.e.d1/p1.e.d2, jar, jv, qg, &it);
    if (b2) p2next = ls_point(d, p2.alpha - p2.e.d1/p2.e.d2, jar, jv, qg, &it);
  }
  return p1.e.cost < p2.e.cost ? p1.alpha : p2.alpha;
}

static double constraint_update(OrcData* d, int nv, const double* jar, const double* qacc, const double* Ma,
                                double* grad, int* nactive_changed, unsigned char* active) {
  double cost = 0; int changed = 0;
  memset(d->qfrc_constraint, 0, sizeof(double)*nv);
  for (int r = 0; r < d->nefc; ++r) {
    unsigned char a = jar[r] < 0;
    if (d->efc_floss[r] > 0) {          /* friction loss: "active" = the quadratic zone (what the Hessian sees); the force acts in all zones */
      double force; int quad;
      cost += row_cost(d->efc_D[r], d->efc_floss[r], jar[r], &force, &quad);
      a = (unsigned char)quad;
      if (a != active[r]) changed = 1;
      active[r] = a;
      d->efc_force[r] = force;
      for (int c = 0; c < nv; ++c) d->qfrc_constraint[c] += d->efc_J[r*nv+c]*force;
      continue;
    }
    if (a != active[r]) changed = 1;
    active[r] = a;
    if (a) { d->efc_force[r] = -d->efc_D[r]*jar[r]; cost += 0.5*d->efc_D[r]*jar[r]*jar[r]; }
    else d->efc_force[r] = 0;
    if (a) { FL(5); for (int c = 0; c < nv; ++c) { d->qfrc_constraint[c] += d->efc_J[r*nv+c]*d->efc_force[r]; if (d->efc_J[r*nv+c] != 0) FL(2); } }
  }
  FL(4*nv + 3*nv);
  double g = 0;
  for (int c = 0; c < nv; ++c) g += (Ma[c]-d->qfrc_smooth[c])*(qacc[c]-d->qacc_smooth[c]);
  cost += 0.5*g;
  for (int c = 0; c < nv; ++c) grad[c] = Ma[c] - d->qfrc_smooth[c] - d->qfrc_constraint[c];
  *nactive_changed = changed;
  return cost;
}

static void newton_solve(const OrcModel* m, OrcData* d) {
  int nv = m->nv, ne = d->nefc;
  double* qacc = d->qacc;
  double *Ma = d->w1, *jar = d->w2, *grad = d->w3, *search = d->w4, *Mv = d->w5;
  double jv[MAXEFC]; unsigned char active[MAXEFC];
  memset(active, 2, sizeof active);
  /* warm start: pick the cheaper of qacc_warmstart and qacc_smooth */
  {
    double costw = 0, costs = 0;
    mul_M(d, nv, Ma, d->qacc_warmstart);
    for (int r = 0; r < ne; ++r) {
      double xw = -d->efc_aref[r], xs = -d->efc_aref[r];
      for (int c = 0; c < nv; ++c) { xw += d->efc_J[r*nv+c]*d->qacc_warmstart[c]; xs += d->efc_J[r*nv+c]*d->qacc_smooth[c]; if (d->efc_J[r*nv+c] != 0) FL(4); }
      FL(8);
      if (d->efc_floss[r] > 0) {
        double force; int quad;
        costw += row_cost(d->efc_D[r], d->efc_floss[r], xw, &force, &quad);
        costs += row_cost(d->efc_D[r], d->efc_floss[r], xs, &force, &quad);
        continue;
      }
      if (xw < 0) costw += 0.5*d->efc_D[r]*xw*xw;
      if (xs < 0) costs += 0.5*d->efc_D[r]*xs*xs;
    }
    double g = 0;
    for (int c = 0; c < nv; ++c) g += (Ma[c]-d->qfrc_smooth[c])*(d->qacc_warmstart[c]-d->qacc_smooth[c]);
    costw += 0.5*g;
    memcpy(qacc, costw < costs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double)*nv);
  }
  mul_M(d, nv, Ma, qacc);
  for (int r = 0; r < ne; ++r) { double s = -d->efc_aref[r]; for (int c = 0; c < nv; ++c) { s += d->efc_J[r*nv+c]*qacc[c]; if (d->efc_J[r*nv+c] != 0) FL(2); } jar[r] = s; }
  int changed;
  double cost = constraint_update(d, nv, jar, qacc, Ma, grad, &changed, active);
  double scale = 1/(m->meaninertia*(nv > 1 ? nv : 1));
  int iter = 0;
  double* H = d->H;
  for (; iter < m->iterations; ) {
    /* Hessian H = M + J' D_active J, Cholesky, search = -H^-1 grad */
    memcpy(H, d->M, sizeof(double)*nv*nv);
    for (int r = 0; r < ne; ++r) if (active[r]) {
      const double* J = d->efc_J + r*nv; double D = d->efc_D[r];
      { int nz = 0; for (int i = 0; i < nv; ++i) if (J[i] != 0) nz++; FL(nz + nz*(nz+1)); }     /* lower triangle of the row's outer product */
      for (int i = 0; i < nv; ++i) if (J[i] != 0) { double t = D*J[i]; for (int j = 0; j < nv; ++j) H[i*nv+j] += t*J[j]; }
    }
    chol_factor(H, H, nv);
    for (int c = 0; c < nv; ++c) search[c] = -grad[c];
    chol_solve(H, search, nv);
    /* exact line search on the piecewise-quadratic cost */
    mul_M(d, nv, Mv, search);
    for (int r = 0; r < ne; ++r) { double s = 0; for (int c = 0; c < nv; ++c) { s += d->efc_J[r*nv+c]*search[c]; if (d->efc_J[r*nv+c] != 0) FL(2); } jv[r] = s; }
    FL(7*nv + 4*nv + 2*ne + 2*nv + 10);
    double qg[3] = {0, 0, 0}, snorm = 0;
    for (int c = 0; c < nv; ++c) {
      qg[1] += search[c]*(Ma[c]-d->qfrc_smooth[c]); qg[2] += 0.5*search[c]*Mv[c]; snorm += search[c]*search[c];
    }
    snorm = sqrt(snorm);
    if (snorm < MINVAL) break;
    double gtol = m->tolerance*LS_TOL*snorm/scale;
    double alpha = 0, lo = 0, hi = -1;
    LsEval e = ls_eval(d, 0, jar, jv, qg);
    if (ls_mode()) alpha = primal_search(d, jar, jv, qg, gtol);
    else
    for (int li = 0; li < LS_ITER; ++li) {
      if (fabs(e.d1) < gtol) break;
      if (e.d1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - e.d1/e.d2;
      if (hi >= 0 && (next <= lo || next >= hi)) next = 0.5*(lo+hi);
      if (next == alpha) break;
      alpha = next;
      e = ls_eval(d, alpha, jar, jv, qg);
    }
    if (alpha == 0) break;
    for (int c = 0; c < nv; ++c) { qacc[c] += alpha*search[c]; Ma[c] += alpha*Mv[c]; }
    for (int r = 0; r < ne; ++r) jar[r] += alpha*jv[r];
    double oldcost = cost;
    cost = constraint_update(d, nv, jar, qacc, Ma, grad, &changed, active);
    iter++;
    double gn = 0; for (int c = 0; c < nv; ++c) gn += grad[c]*grad[c];
    double improvement = scale*(oldcost-cost), gradient = scale*sqrt(gn);
    if (improvement < m->tolerance || gradient < m->tolerance) break;
  }
  d->solver_iter = iter;
}

static void fwd_acceleration(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  FL(2*nv);
  for (int c = 0; c < nv; ++c) d->qfrc_smooth[c] = d->qfrc_passive[c] - d->qfrc_bias[c] + d->qfrc_actuator[c];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double)*nv);
  chol_solve(d->Mchol, d->qacc_smooth, nv);
  if (d->nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
    memset(d->qfrc_constraint, 0, sizeof(double)*nv);
    d->solver_iter = 0;
    return;
  }
  STAGE(ST_NEWTON); newton_solve(m, d); STAGE(ST_ACCEL);
}

void orc_fwd_position(const OrcModel* m, OrcData* d) {
  STAGE(ST_KIN); orc_kinematics(m, d);
  STAGE(ST_COM); com_pos(m, d);
  STAGE(ST_TENDON); tendon(m, d); transmission(m, d);
  STAGE(ST_CRB); crb(m, d);
  STAGE(ST_COLLIDE); collision(m, d);
  STAGE(ST_CONSTRAINT); make_constraint(m, d);
  STAGE(ST_TASK);
}

void orc_forward(const OrcModel* m, OrcData* d) {
  orc_fwd_position(m, d);
  STAGE(ST_VELOCITY); fwd_velocity(m, d);
  STAGE(ST_ACTUATION); fwd_actuation(m, d);
  STAGE(ST_ACCEL); fwd_acceleration(m, d);
  STAGE(ST_TASK);
}

/* ------------------------------------------------------------------ P11 integrators */
static void integrate_pos(const OrcModel* m, double* qpos, const double* qvel, double h) {
  for (int j = 0; j < m->njnt; ++j) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MYO_JNT_FREE) {
      for (int k = 0; k < 3; ++k) qpos[qa+k] += h*qvel[da+k];
      double w[3] = { qvel[da+3], qvel[da+4], qvel[da+5] };
      double ang = h*norm3(w);
      if (ang > 0) {
        double ax[3] = { w[0], w[1], w[2] }, qr[4];
        normalize3(ax);
        axisangle2quat(qr, ax, ang);
        mulquat(qpos+qa+3, qpos+qa+3, qr);
      }
      normalize4(qpos+qa+3);
    } else qpos[qa] += h*qvel[da];
  }
}

static void advance(const OrcModel* m, OrcData* d, const double* act_dot, const double* qacc, const double* qvel_for_pos) {
  double h = m->timestep;
  FL(2*m->na + 2*m->nv + 2*m->nq + 1);
  for (int i = 0; i < m->na; ++i) {
    d->act[i] += h*act_dot[i];
    int iu = i + (m->nu - m->na);
    if (m->actuator_dyntype[iu] == MYO_DYN_MUSCLE) { if (d->act[i] < 0) d->act[i] = 0; else if (d->act[i] > 1) d->act[i] = 1; }
  }
  for (int c = 0; c < m->nv; ++c) d->qvel[c] += h*qacc[c];
  integrate_pos(m, d->qpos, qvel_for_pos ? qvel_for_pos : d->qvel, h);
  d->time += h;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double)*m->nv);
}

static void check_state(const OrcModel* m, OrcData* d) {
  for (int i = 0; i < m->nq; ++i) if (!isfinite(d->qpos[i]) || fabs(d->qpos[i]) > 1e10) d->bad = 1;
  for (int i = 0; i < m->nv; ++i) if (!isfinite(d->qvel[i]) || fabs(d->qvel[i]) > 1e10) d->bad = 1;
}

static void euler(const OrcModel* m, OrcData* d) {
  int nv = m->nv; int damped = 0;
  for (int c = 0; c < nv; ++c) if (m->dof_damping[c] > 0) damped = 1;
  if (!damped) { advance(m, d, d->act_dot, d->qacc, NULL); return; }
  double* H = d->H; double* qa = d->w1;
  memcpy(H, d->M, sizeof(double)*nv*nv);
  FL(3*nv);
  for (int c = 0; c < nv; ++c) { H[c*nv+c] += m->timestep*m->dof_damping[c]; qa[c] = d->qfrc_smooth[c] + d->qfrc_constraint[c]; }
  chol_factor(H, H, nv);
  chol_solve(H, qa, nv);
  advance(m, d, d->act_dot, qa, NULL);
}

static void rk4(const OrcModel* m, OrcData* d) {
  static const double A[9] = {0.5,0,0, 0,0.5,0, 0,0,1}, B[4] = {1.0/6,1.0/3,1.0/3,1.0/6};
  int nq = m->nq, nv = m->nv, na = m->na; double h = m->timestep, t0 = d->time;
  double *X0q = (double*)malloc(sizeof(double)*(nq+nv+na)), *F[4];
  double *X0v = X0q+nq, *X0a = X0v+nv;
  memcpy(X0q, d->qpos, sizeof(double)*nq); memcpy(X0v, d->qvel, sizeof(double)*nv); memcpy(X0a, d->act, sizeof(double)*na);
  for (int i = 0; i < 4; ++i) F[i] = (double*)malloc(sizeof(double)*(2*nv+na));
  double* dX = (double*)malloc(sizeof(double)*(2*nv+na));
  memcpy(F[0], d->qvel, sizeof(double)*nv); memcpy(F[0]+nv, d->qacc, sizeof(double)*nv); memcpy(F[0]+2*nv, d->act_dot, sizeof(double)*na);
  for (int i = 1; i < 4; ++i) {
    memset(dX, 0, sizeof(double)*(2*nv+na));
    for (int j = 0; j < i; ++j) for (int k = 0; k < 2*nv+na; ++k) dX[k] += A[(i-1)*3+j]*F[j][k];
    memcpy(d->qpos, X0q, sizeof(double)*nq);
    integrate_pos(m, d->qpos, dX, h);
    for (int k = 0; k < nv; ++k) d->qvel[k] = X0v[k] + h*dX[nv+k];
    for (int k = 0; k < na; ++k) d->act[k] = X0a[k] + h*dX[2*nv+k];
    double c = 0; for (int j = 0; j < i; ++j) c += A[(i-1)*3+j];
    d->time = t0 + h*c;
    orc_forward(m, d); STAGE(ST_INTEGRATE); FL(2*(2*nv+na)*i + 2*(nv+na));
    memcpy(F[i], d->qvel, sizeof(double)*nv); memcpy(F[i]+nv, d->qacc, sizeof(double)*nv); memcpy(F[i]+2*nv, d->act_dot, sizeof(double)*na);
  }
  memset(dX, 0, sizeof(double)*(2*nv+na));
  for (int j = 0; j < 4; ++j) for (int k = 0; k < 2*nv+na; ++k) dX[k] += B[j]*F[j][k];
  memcpy(d->qpos, X0q, sizeof(double)*nq); memcpy(d->qvel, X0v, sizeof(double)*nv); memcpy(d->act, X0a, sizeof(double)*na);
  d->time = t0;
  advance(m, d, dX+2*nv, dX+nv, dX);
  for (int i = 0; i < 4; ++i) free(F[i]);
  free(dX); free(X0q);
}

void orc_step(const OrcModel* m, OrcData* d) {
  check_state(m, d);
  orc_forward(m, d);
  for (int c = 0; c < m->nv; ++c) if (!isfinite(d->qacc[c]) || fabs(d->qacc[c]) > 1e10) d->bad = 1;
  STAGE(ST_INTEGRATE);
  if (m->integrator == MYO_INT_RK4) rk4(m, d); else euler(m, d);
  STAGE(ST_TASK);
}

/* ------------------------------------------------------------------ Baoding task layer */
void orc_baoding_obs(const OrcModel* m, OrcData* d, const OrcBaodingCfg* cfg, double* obs) {
  /* layout: SURVEY.md §8a-T1 (pinned by tests/golden/reset_obs_golden.npy, obs_snapshots.npz) */
  double dt = cfg->frame_skip*m->timestep;
  int nh = cfg->n_hand, o = 0;
  for (int i = 0; i < nh; ++i) obs[o++] = d->qpos[i];
  int nv = m->nv;
  const double* s1 = d->site_xpos + 3*cfg->obj1_sid; const double* s2 = d->site_xpos + 3*cfg->obj2_sid;
  const double* t1 = d->site_xpos + 3*cfg->target1_sid; const double* t2 = d->site_xpos + 3*cfg->target2_sid;
  for (int k = 0; k < 3; ++k) obs[o++] = s1[k];
  for (int k = 0; k < 3; ++k) obs[o++] = d->qvel[nv-12+k]*dt;
  for (int k = 0; k < 3; ++k) obs[o++] = s2[k];
  for (int k = 0; k < 3; ++k) obs[o++] = d->qvel[nv-6+k]*dt;
  for (int k = 0; k < 3; ++k) obs[o++] = t1[k];
  for (int k = 0; k < 3; ++k) obs[o++] = t2[k];
  for (int k = 0; k < 3; ++k) obs[o++] = t1[k]-s1[k];
  for (int k = 0; k < 3; ++k) obs[o++] = t2[k]-s2[k];
  for (int i = 0; i < m->na; ++i) obs[o++] = d->act[i];
}

void orc_baoding_reward(const OrcBaodingCfg* cfg, int na, const double* obs, double* c) {
  /* restates /root/reference/src/envs/baoding.py:24-94 (P1) == :403-467 (P2) */
  int nh = cfg->n_hand;
  const double* e1 = obs + nh + 18; const double* e2 = obs + nh + 21; const double* act = obs + nh + 24;
  double d1 = sqrt(e1[0]*e1[0]+e1[1]*e1[1]+e1[2]*e1[2]), d2 = sqrt(e2[0]*e2[0]+e2[1]*e2[1]+e2[2]*e2[2]);
  double am = 0; for (int i = 0; i < na; ++i) am += act[i]*act[i];
  am = na ? sqrt(am)/na : 0;
  int fall = (obs[nh+2] < cfg->drop_th) || (obs[nh+8] < cfg->drop_th);
  c[0] = -d1; c[1] = -d2; c[2] = -am; c[3] = fall ? 0 : 1; c[4] = -(d1+d2);
  c[5] = (d1 < cfg->proximity_th) && (d2 < cfg->proximity_th) && !fall; c[6] = fall;
  double dense = 0; for (int k = 0; k < 7; ++k) dense += cfg->w[k]*c[k];
  c[7] = dense;
}

void orc_baoding_step(const OrcModel* m, OrcData* d, const OrcBaodingCfg* cfg, OrcBaodingState* st,
                      const float* action, double* obs, double* comps) {
  /* BaodingEnvV1.step [3P-RECALL, SURVEY §3.3]: place targets from the goal schedule, then BaseV0.step */
  double dt = cfg->frame_skip*m->timestep;
  if (st->which_task != 0) {
    double sign = st->which_task == 1 ? -1.0 : 1.0;
    double ang = sign*2*M_PI*(st->counter*dt/st->time_period);
    double a1 = ang + st->start_angle[0], a2 = ang + st->start_angle[1];
    d->site_pos[3*cfg->target1_sid]   = st->x_radius*cos(a1) + cfg->center_pos[0];
    d->site_pos[3*cfg->target1_sid+1] = st->y_radius*sin(a1) + cfg->center_pos[1];
    d->site_pos[3*cfg->target2_sid]   = st->x_radius*cos(a2) + cfg->center_pos[0];
    d->site_pos[3*cfg->target2_sid+1] = st->y_radius*sin(a2) + cfg->center_pos[1];
  }
  st->counter++;
  for (int i = 0; i < m->nu; ++i) { /* normalize_act: float32 sigmoid(5(a-0.5)) after clip */
    float a = action[i]; if (a < -1.f) a = -1.f; else if (a > 1.f) a = 1.f;
    /* every operation but exp is an IEEE float32 operation (identical on any platform); exp is taken as the
     * CORRECTLY ROUNDED float32 exponential, (float)exp((double)x) — numpy's float32 exp, glibc's expf and a GPU's
     * expf each differ from it by at most 1 ulp, and from each other, on some arguments */
    float c = 1.0f/(1.0f + (float)exp((double)(-5.0f*(a-0.5f))));
    d->ctrl[i] = (double)c;
  }
  for (int k = 0; k < cfg->frame_skip; ++k) orc_step(m, d);
  orc_kinematics(m, d);
  orc_baoding_obs(m, d, cfg, obs);
  orc_baoding_reward(cfg, m->na, obs, comps);
}

/* ------------------------------------------------------------------ die-reorient task layer
 * CustomReorientEnv (/root/reference/src/envs/reorient.py) over ReorientEnvV0 (MyoSuite 1.2.3, not in /root/reference:
 * observation layout and the Euler-angle helpers are the published ones [3P-RECALL]). */
static void ro_mat2euler(const double* m, double* e) {
  /* myosuite.utils.quat_math.mat2euler == mujoco-py rotations.mat2euler; _EPS4 = 4 * float64 eps */
  double cy = sqrt(m[8]*m[8] + m[5]*m[5]);
  int cond = cy > 8.881784197001252e-16;
  e[2] = cond ? -atan2(m[1], m[0]) : -atan2(-m[3], m[4]);
  e[1] = -atan2(-m[2], cy);
  e[0] = cond ? -atan2(m[5], m[8]) : 0.0;
}
void orc_euler2quat(const double* e, double* q) {
  /* myosuite.utils.quat_math.euler2quat (reorient.py:203-205 calls it for the goal orientation) */
  double ai = e[2]/2, aj = -e[1]/2, ak = e[0]/2;
  double si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
  double cc = ci*ck, cs = ci*sk, sc = si*ck, ss = si*sk;
  q[0] = cj*cc + sj*ss; q[1] = cj*cs - sj*sc; q[2] = -(cj*ss + sj*cc); q[3] = cj*sc - sj*cs;
}

void orc_reorient_set_die(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, const double* friction, double del_size) {
  /* reorient.py:136-147: per-geom friction; capsule half-lengths (size[:,1]) of all but the last three die geoms and all
   * three sizes of the last three grow by del_size; every geom centre moves outward by it (pos/|pos + 1e-16| * (|pos0| + del)).
   * geom_rbound is not refreshed (no mj_setConst), as in the Baoding P2 reset. */
  int g0 = cfg->gid0, gn = cfg->gidn;
  if (friction) memcpy(d->geom_friction + 3*g0, friction, sizeof(double)*3*(gn-g0));
  for (int g = g0; g < gn; ++g) {
    if (g < gn-3) d->geom_size[3*g+1] = m->geom_size[3*g+1] + del_size;
    else for (int k = 0; k < 3; ++k) d->geom_size[3*g+k] = m->geom_size[3*g+k] + del_size;
    for (int k = 0; k < 3; ++k) {
      double p = m->geom_pos[3*g+k];
      d->geom_pos[3*g+k] = p/fabs(p + 1e-16)*(fabs(p) + del_size);
    }
  }
}

void orc_reorient_obs(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, const OrcReorientState* st, double* obs) {
  /* hand_qpos, hand_qvel*dt, obj_pos, goal_pos, pos_err, obj_rot, goal_rot, rot_err, act.  The goal body hangs off the world
   * with the episode's pose (reset writes model.body_pos / body_quat of `target`, reorient.py:125-129,203-205); both sites
   * are taken with identity local orientation (site_xmat = xmat of the body). */
  double dt = cfg->frame_skip*m->timestep;
  int nh = cfg->n_hand, o = 0;
  for (int i = 0; i < nh; ++i) obs[o++] = d->qpos[i];
  for (int i = 0; i < nh; ++i) obs[o++] = d->qvel[i]*dt;
  const double* op = d->site_xpos + 3*cfg->object_sid;
  double gq[4], gm[9], gp[3], t[3], oe[3], ge[3];
  double n = sqrt(st->goal_quat[0]*st->goal_quat[0] + st->goal_quat[1]*st->goal_quat[1] + st->goal_quat[2]*st->goal_quat[2] + st->goal_quat[3]*st->goal_quat[3]);
  for (int k = 0; k < 4; ++k) gq[k] = st->goal_quat[k]/n;
  quat2mat(gm, gq);
  mulmatvec3(t, gm, m->site_pos + 3*cfg->goal_sid);
  for (int k = 0; k < 3; ++k) gp[k] = st->goal_pos[k] + t[k];
  ro_mat2euler(d->xmat + 9*cfg->object_bid, oe);
  ro_mat2euler(gm, ge);
  for (int k = 0; k < 3; ++k) obs[o++] = op[k];
  for (int k = 0; k < 3; ++k) obs[o++] = gp[k];
  for (int k = 0; k < 3; ++k) obs[o++] = gp[k] - op[k] - cfg->goal_obj_offset[k];
  for (int k = 0; k < 3; ++k) obs[o++] = oe[k];
  for (int k = 0; k < 3; ++k) obs[o++] = ge[k];
  for (int k = 0; k < 3; ++k) obs[o++] = ge[k] - oe[k];
  for (int i = 0; i < m->na; ++i) obs[o++] = d->act[i];
}

void orc_reorient_reward(const OrcReorientCfg* cfg, int na, const double* pos_err, const double* rot_err, const double* act,
                         double prev_pos_dist, double prev_rot_dist, double* c) {
  /* restates /root/reference/src/envs/reorient.py:12-56; c = pos_dist, rot_dist, pos_dist_diff, rot_dist_diff, alive,
   * act_reg, sparse, solved, done, dense (the OrderedDict's order) */
  double pd = fabs(sqrt(pos_err[0]*pos_err[0] + pos_err[1]*pos_err[1] + pos_err[2]*pos_err[2]));
  double rd = fabs(sqrt(rot_err[0]*rot_err[0] + rot_err[1]*rot_err[1] + rot_err[2]*rot_err[2]));
  double am = 0; for (int i = 0; i < na; ++i) am += act[i]*act[i];
  am = na ? sqrt(am)/na : 0;
  int drop = pd > cfg->drop_th;
  c[0] = -1.0*pd; c[1] = -1.0*rd; c[2] = prev_pos_dist - pd; c[3] = prev_rot_dist - rd; c[4] = !drop; c[5] = -1.0*am;
  c[6] = -rd - 10.0*pd; c[7] = (pd < cfg->pos_th) && (rd < cfg->rot_th) && !drop; c[8] = drop;
  double dense = 0; for (int k = 0; k < 9; ++k) dense += cfg->w[k]*c[k];
  c[9] = dense;
}

void orc_reorient_step(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, OrcReorientState* st, const float* action,
                       double* obs, double* comps) {
  /* BaseV0.step (clip, float32 sigmoid for the muscles, frame_skip physics steps, obs, reward) followed by
   * CustomReorientEnv.step's update of the shaping distances (reorient.py:207-212) */
  for (int i = 0; i < m->nu; ++i) {
    float a = action[i]; if (a < -1.f) a = -1.f; else if (a > 1.f) a = 1.f;
    float c = 1.0f/(1.0f + (float)exp((double)(-5.0f*(a-0.5f))));     /* see orc_baoding_step */
    d->ctrl[i] = (double)c;
  }
  for (int k = 0; k < cfg->frame_skip; ++k) orc_step(m, d);
  orc_kinematics(m, d);
  orc_reorient_obs(m, d, cfg, st, obs);
  int o = 2*cfg->n_hand;
  orc_reorient_reward(cfg, m->na, obs + o + 6, obs + o + 15, obs + o + 18, st->pos_dist, st->rot_dist, comps);
  st->pos_dist = -comps[0]; st->rot_dist = -comps[1];
}

void orc_reorient_reset_dists(const OrcModel* m, OrcData* d, const OrcReorientCfg* cfg, OrcReorientState* st, double* obs) {
  /* tail of CustomReorientEnv.reset (reorient.py:177-180): observation of the reset state and its two distances */
  orc_kinematics(m, d);
  orc_reorient_obs(m, d, cfg, st, obs);
  const double* pe = obs + 2*cfg->n_hand + 6; const double* re = obs + 2*cfg->n_hand + 15;
  st->pos_dist = fabs(sqrt(pe[0]*pe[0] + pe[1]*pe[1] + pe[2]*pe[2]));
  st->rot_dist = fabs(sqrt(re[0]*re[0] + re[1]*re[1] + re[2]*re[2]));
}
