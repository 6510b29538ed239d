// myo_sparse_ldl.h — tree-sparse L'DL factorisation and solve of the joint-space inertia matrix, level-parallel.
//
// What MuJoCo's mj_factorM / mj_solveM do (restated scalar in oracle/myo_oracle.c: orc_factorM / orc_solveM; SURVEY.md §8a
// P5, P10, P11): M has the sparsity of the kinematic tree — row i holds (i,i), (i,parent), (i,grandparent), ... in
// dof_Madr order — and eliminating dofs from the leaves towards the roots creates no fill-in.  The two M-only systems of
// a substep (qacc_smooth = M^-1 qfrc_smooth; Euler's implicit-damping solve (M + h diag(b)) qacc' = f) went through the
// DENSE Cholesky of the Newton step before: 24-36 serial column steps for a matrix that is five 4-dof chains on a
// 3-dof wrist.  Here the serial depth is the DEPTH of the dof tree (7 for the hand):
//
//   factor  rows k of one depth are final once every deeper row has been eliminated; each (k; i >= j among k's strict
//           ancestors) is one work item  M[i,j] -= M[k,i] M[k,j] / M[k,k]  (rows of different branches hit the same
//           wrist entries: LDS adds).  Deepest level first, one phase per 64 items.
//   solve   x <- L^-T x (deepest level first; item (i,j), j a strict ancestor of i:  x[j] -= L[i,j] x[i]),
//           x <- D^-1 x, x <- L^-1 x (shallowest level first, the same items:  x[i] -= L[i,j] x[j]).
//
// The item lists are host-built tables (myobatch.hip: build_ldl_tables), padded to whole 64-item chunks per level with
// no-op items; a lane fetches all its words before the first phase (one table latency per call, not one per level).
// Rows known to be diagonal (dofs >= nlead: free bodies whose inertial frame is the body frame) carry no items.
// Models whose tree needs more than MYO_LD_FQ / MYO_LD_SQ chunks keep the dense path (ld_nfq == 0).
#pragma once

#define MYO_LD_NOP 255
static_assert(MYO_NM_MAX < MYO_LD_NOP, "8-bit entry indices in the item words");
static_assert(MYO_NM_MAX + MYO_NV_MAX <= MYO_H_SIZE, "the factor (nM entries) and 1/D (nv) live in H");

// x <- (M + diag_scale * diag(dof_damping))^-1 x ; qLD / 1/D are left in H (H is dead at both call sites)
template <typename T, int NC>
DEVFN void ldl_factor_solve(const DevModel<T>& M_in, Scratch<T, NC>& s_in, LREF(T) x_r, int damped) {
  MYO_BIND_M(T) MYO_BIND_S(T)
  WAVE_FN
  T* const x = LPTR(T, x_r);
  T* const qLD = s.H;
  T* const dinv = s.H + MYO_NM_MAX;
  // every table word of this lane, requested up front
  LANE_VAR(int, fw0); LANE_VAR(int, fw1); LANE_VAR(int, fw2); LANE_VAR(int, fw3); LANE_VAR(int, fw4); LANE_VAR(int, fw5);
  LANE_VAR(int, fw6); LANE_VAR(int, fw7); LANE_VAR(int, fw8); LANE_VAR(int, fw9); LANE_VAR(int, fw10); LANE_VAR(int, fw11);
  LANE_VAR(int, sw0); LANE_VAR(int, sw1); LANE_VAR(int, sw2); LANE_VAR(int, sw3); LANE_VAR(int, sw4); LANE_VAR(int, sw5);
  LANE_VAR(int, sw6); LANE_VAR(int, sw7);
  static_assert(MYO_LD_FQ == 12 && MYO_LD_SQ == 8, "one lane variable per chunk");
  PHASE {
    LV(fw0) = M.ld_fac[lane]; LV(fw1) = M.ld_fac[64 + lane]; LV(fw2) = M.ld_fac[128 + lane]; LV(fw3) = M.ld_fac[192 + lane];
    LV(fw4) = M.ld_fac[256 + lane]; LV(fw5) = M.ld_fac[320 + lane]; LV(fw6) = M.ld_fac[384 + lane]; LV(fw7) = M.ld_fac[448 + lane];
    LV(fw8) = M.ld_fac[512 + lane]; LV(fw9) = M.ld_fac[576 + lane]; LV(fw10) = M.ld_fac[640 + lane]; LV(fw11) = M.ld_fac[704 + lane];
    LV(sw0) = M.ld_sol[lane]; LV(sw1) = M.ld_sol[64 + lane]; LV(sw2) = M.ld_sol[128 + lane]; LV(sw3) = M.ld_sol[192 + lane];
    LV(sw4) = M.ld_sol[256 + lane]; LV(sw5) = M.ld_sol[320 + lane]; LV(sw6) = M.ld_sol[384 + lane]; LV(sw7) = M.ld_sol[448 + lane];
    // qLD <- M (+ h b on the diagonal)
    constexpr int NE = (MYO_NM_MAX + 63) / 64;
    int pk[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) pk[q] = M.M_pk[(lane + 64 * q) < MYO_NM_MAX ? (lane + 64 * q) : 0];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      if (e < M.nM) {
        const int i = pk[q] & 255, j = (pk[q] >> 8) & 255;
        T v = s.qM[e];
        if (damped && i == j) v += M.timestep * M.dof_damping[i];
        qLD[e] = v;
      }
    }
  }
  SYNC();
  const int nfq = M.ld_nfq, nsq = M.ld_nsq;
#define MYO_LD_FAC(q, w)                                                                  \
  if ((q) < nfq) {                                                                         \
    PHASE {                                                                                \
      const int e_ij = LV(w) & 255, e_ki = (LV(w) >> 8) & 255, e_kj = (LV(w) >> 16) & 255, e_kk = (LV(w) >> 24) & 255; \
      if (e_ij != MYO_LD_NOP) {                                                            \
        const T tmp = qLD[e_ki] / qLD[e_kk];                                               \
        lds_add(&qLD[e_ij], -(qLD[e_kj] * tmp));                                           \
      }                                                                                    \
    }                                                                                      \
    SYNC();                                                                                \
  }
  MYO_LD_FAC(0, fw0) MYO_LD_FAC(1, fw1) MYO_LD_FAC(2, fw2) MYO_LD_FAC(3, fw3) MYO_LD_FAC(4, fw4) MYO_LD_FAC(5, fw5)
  MYO_LD_FAC(6, fw6) MYO_LD_FAC(7, fw7) MYO_LD_FAC(8, fw8) MYO_LD_FAC(9, fw9) MYO_LD_FAC(10, fw10) MYO_LD_FAC(11, fw11)
#undef MYO_LD_FAC
  PHASE {
    const int i = lane;
    if (i < M.nv) dinv[i] = (T)1 / qLD[M.dof_Madr[i]];
  }
  SYNC();
  // x <- L^-T x : deepest rows first
#define MYO_LD_BWD(q, w)                                                                  \
  if ((q) < nsq) {                                                                         \
    PHASE {                                                                                \
      const int e_ij = LV(w) & 255, i = (LV(w) >> 8) & 255, j = (LV(w) >> 16) & 255;       \
      if (e_ij != MYO_LD_NOP) lds_add(&x[j], -(qLD[e_ij] * (x[i] * dinv[i])));             \
    }                                                                                      \
    SYNC();                                                                                \
  }
  MYO_LD_BWD(0, sw0) MYO_LD_BWD(1, sw1) MYO_LD_BWD(2, sw2) MYO_LD_BWD(3, sw3) MYO_LD_BWD(4, sw4) MYO_LD_BWD(5, sw5)
  MYO_LD_BWD(6, sw6) MYO_LD_BWD(7, sw7)
#undef MYO_LD_BWD
  // x <- D^-1 x
  PHASE {
    const int i = lane;
    if (i < M.nv) x[i] = x[i] * dinv[i];
  }
  SYNC();
  // x <- L^-1 x : shallowest rows first (the chunks in reverse)
#define MYO_LD_FWD(q, w)                                                                  \
  if ((q) < nsq) {                                                                         \
    PHASE {                                                                                \
      const int e_ij = LV(w) & 255, i = (LV(w) >> 8) & 255, j = (LV(w) >> 16) & 255;       \
      if (e_ij != MYO_LD_NOP) lds_add(&x[i], -((qLD[e_ij] * dinv[i]) * x[j]));             \
    }                                                                                      \
    SYNC();                                                                                \
  }
  MYO_LD_FWD(7, sw7) MYO_LD_FWD(6, sw6) MYO_LD_FWD(5, sw5) MYO_LD_FWD(4, sw4) MYO_LD_FWD(3, sw3) MYO_LD_FWD(2, sw2)
  MYO_LD_FWD(1, sw1) MYO_LD_FWD(0, sw0)
#undef MYO_LD_FWD
}
