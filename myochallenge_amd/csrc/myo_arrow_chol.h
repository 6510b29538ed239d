// myo_arrow_chol.h — Newton's linear system  (M + J'DJ) x = b  solved in its BLOCK-ARROW structure (gfx950 build only).
//
// mj_solNewton factors the dense nv x nv Hessian with a Cholesky (restated dense in oracle/myo_oracle.c).  For
// the MyoHand the matrix is not dense: the five fingers are 4-dof chains that no constraint couples to each other — M couples a
// dof with its ancestors and descendants, a tendon-limit row the dofs of one finger's tendon (+ wrist), a contact the dofs of one
// finger link (+ wrist) and of a ball — so with the rows ordered [separator S | finger 0 | ... | finger 4] (host table hperm,
// myobatch.hip: build_arrow_tables; S = wrist + free bodies, <= 16 rows) H is
//
//        | H_SS   H_S0 ... H_S4 |          the finger blocks are eliminated FIRST and all at once:
//        | H_0S   H_00          |            L_ff L_ff' = H_ff                        (4 x 4, one finger per 16-lane group)
//        |  ...         ...     |            L_Sf = H_Sf L_ff^-T                      (row r of S = lane & 15)
//        | H_4S            H_44 |            H_SS <- H_SS - sum_f L_Sf L_Sf'          (one 16x16x4 MFMA per finger into ONE tile)
//                                           then the 16 x 16 Schur complement goes through the dense kernel (chol_factor_solve_reg<T, 16>)
//
// and the substitutions follow the same order (fingers forward, separator forward + backward, fingers backward).  The serial
// chain is 4 + 16 pivots instead of 36, the fp64 factorisation issues 5 + 3 MFMAs instead of 31 and fills one accumulator tile
// instead of six; measured in DESIGN.md §5.  Same arithmetic per entry as the dense factorisation up to the order of the
// eliminations (a different, equally valid Cholesky ordering): parity with the oracle is checked at the stepper's tolerance.
#pragma once
#ifndef MYO_EMU

template <typename T> __device__ __forceinline__ T myo_row16_sum(T v);
template <> __device__ __forceinline__ float myo_row16_sum<float>(float v) {
  v += myo_dpp_f(v, 0); v += myo_dpp_f(v, 1); v += myo_dpp_f(v, 2); v += myo_dpp_f(v, 3);
  return v;                                            // every lane of a 16-lane row holds the row's total
}
template <> __device__ __forceinline__ double myo_row16_sum<double>(double v) {
  v += myo_dpp_d(v, 0); v += myo_dpp_d(v, 1); v += myo_dpp_d(v, 2); v += myo_dpp_d(v, 3);
  return v;
}

// packed-H offset of row 16 + 4 f + t (myo_hrow with q = 4 + f)
__device__ __forceinline__ int myo_arrow_row(int f, int t) { const int q = 4 + f; return ((q * (q + 1)) << 3) + ((t * (q + 1)) << 2); }

// x (dof order, LDS) <- H^-1 x ; H in s.H in hperm order (load_H_from_M(perm = 1) + build_hessian), destroyed.  Three leaf
// calls from kernel level (no function of the stepper calls another one):  arrow_eliminate_blocks(x_r);
// chol_factor_solve_reg<T, 16, NC>(offset of s.Mv, 16);  arrow_finish(x_r).  The right-hand side travels in row order in s.Mv.
template <typename T, int NC>
__device__ __noinline__ void arrow_eliminate_blocks(int x_r) {
  typedef MyoMfma<T> MM;
  typedef __attribute__((address_space(3))) T* lds_t;
  typedef __attribute__((address_space(3))) const unsigned char* lds_b;
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  const DevModel<T>& M = myo_cmodel<T>();
  const int lane = threadIdx.x, lc = lane & 15, lq = lane >> 4;
  const int nf = M.arrow_nf, nv = M.nv;
  static_assert(MYO_ARROW_S == 16 && MYO_ARROW_B == 4 && MYO_ARROW_NF == 5, "one MFMA tile of separator rows, blocks of K = 4 rows, a fifth block shared by all lane groups");
  static_assert(4 * 4 * 16 <= MYO_NB_MAX * 6 + 2 * (MYO_NLIM_MAX + 4 * NC), "operand stage in bvec + efc_jv + efc_force");
  lds_t Hp = (lds_t)s.H, xp = (lds_t)s.Mv, stage = (lds_t)S_SOLVE_STAGE(s), xin = (lds_t)LPTR(T, x_r);
  lds_b perm = (lds_b)s.hperm;
  asm volatile("" : "+v"(Hp), "+v"(xp), "+v"(stage), "+v"(xin), "+v"(perm));
  // ---- right-hand side in row order (rows without a dof: 0)
  const int myrow = lane < nv ? (int)perm[lane] : 0;
  {
    const T v = lane < nv ? xin[lane < nv ? lane : 0] : (T)0;
    if (lane < nv) xp[myrow] = v;
    if (lane < MYO_NV_MAX && ((M.arrow_pad >> lane) & 1ull)) xp[lane] = 0;
  }
  SYNC();
  // ---- finger blocks.  Lane group g = lane >> 4 works on finger g (groups >= nf shadow the last finger and store nothing);
  // a fifth finger is done by all four groups redundantly (so its MFMA operand needs no exchange), group 0 stores.
  T y[2][4];                                            // L_Sf[r = lc][t], round 0: finger g, round 1: finger 4
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (round == 1 && nf <= 4) break;
    const int f = round ? 4 : (lq < nf ? lq : nf - 1);
    const bool writer = round ? (lq == 0) : (lq < nf);
    const int r0 = myo_arrow_row(f, 0), r1 = myo_arrow_row(f, 1), r2 = myo_arrow_row(f, 2), r3 = myo_arrow_row(f, 3);
    const int c0 = MYO_ARROW_S + 4 * f;
    // everything this round reads, before anything it writes
    T d00 = Hp[r0 + c0];
    const T d10 = Hp[r1 + c0]; T d11 = Hp[r1 + c0 + 1];
    const T d20 = Hp[r2 + c0], d21 = Hp[r2 + c0 + 1]; T d22 = Hp[r2 + c0 + 2];
    const T d30 = Hp[r3 + c0], d31 = Hp[r3 + c0 + 1], d32 = Hp[r3 + c0 + 2]; T d33 = Hp[r3 + c0 + 3];
    const T a0 = Hp[r0 + lc], a1 = Hp[r1 + lc], a2 = Hp[r2 + lc], a3 = Hp[r3 + lc];
    const T b0 = xp[c0], b1 = xp[c0 + 1], b2 = xp[c0 + 2], b3 = xp[c0 + 3];
    // 4 x 4 Cholesky (the arithmetic of the dense kernel's diagonal block)
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
    // row lc of L_Sf and the block's part of the forward substitution: the same recurrence  L_ff w = a
    const T y0 = a0 * i0, y1 = (a1 - y0 * l10) * i1, y2 = (a2 - y0 * l20 - y1 * l21) * i2, y3 = (a3 - y0 * l30 - y1 * l31 - y2 * l32) * i3;
    const T v0 = b0 * i0, v1 = (b1 - v0 * l10) * i1, v2 = (b2 - v0 * l20 - v1 * l21) * i2, v3 = (b3 - v0 * l30 - v1 * l31 - v2 * l32) * i3;
    y[round][0] = y0; y[round][1] = y1; y[round][2] = y2; y[round][3] = y3;
    // b_S -= L_Sf y_f  (row lc; up to five lanes add into one slot)
    if (writer) {
      lds_add((T*)(xp + lc), -(y0 * v0 + y1 * v1 + y2 * v2 + y3 * v3));
      Hp[r0 + lc] = y0; Hp[r1 + lc] = y1; Hp[r2 + lc] = y2; Hp[r3 + lc] = y3;       // L_Sf replaces H_fS (row-of-finger storage)
      if (round == 0) { stage[(4 * lq + 0) * 16 + lc] = y0; stage[(4 * lq + 1) * 16 + lc] = y1; stage[(4 * lq + 2) * 16 + lc] = y2; stage[(4 * lq + 3) * 16 + lc] = y3; }
      if (lc == 0) {
        // L_ff (strictly lower part) and the INVERSE diagonal stay in the block's slots for the backward substitution; y_f in xp
        Hp[r0 + c0] = i0;
        Hp[r1 + c0] = l10; Hp[r1 + c0 + 1] = i1;
        Hp[r2 + c0] = l20; Hp[r2 + c0 + 1] = l21; Hp[r2 + c0 + 2] = i2;
        Hp[r3 + c0] = l30; Hp[r3 + c0 + 1] = l31; Hp[r3 + c0 + 2] = l32; Hp[r3 + c0 + 3] = i3;
        xp[c0] = v0; xp[c0 + 1] = v1; xp[c0 + 2] = v2; xp[c0 + 3] = v3;
      }
    }
  }
  // ---- Schur complement of the separator on the matrix cores: one tile, one MFMA per finger
  typename MM::V4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = MM::crow(lane, r), j = lc;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    acc[r] = Hp[MYO_HIDX(hi, lo)];
  }
  SYNC();
  {
    const int nq = nf < 4 ? nf : 4;
    T op[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) op[f] = stage[(4 * (f < nq ? f : 0) + lq) * 16 + lc];
#pragma unroll
    for (int f = 0; f < 4; ++f) if (f < nq) acc = MM::mma(-op[f], op[f], acc);
    if (nf > 4) {
      const T o4 = lq == 0 ? y[1][0] : (lq == 1 ? y[1][1] : (lq == 2 ? y[1][2] : y[1][3]));
      acc = MM::mma(-o4, o4, acc);
    }
  }
  // lower triangle back to H rows 0..15 (the other lanes store into the stage, which has been read)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = MM::crow(lane, r), j = lc;
    lds_t dst = i >= j ? Hp + MYO_HIDX(i, j) : stage + lane;
    *dst = acc[r];
  }
  SYNC();
}

// fingers backward:  x_f = L_ff^-T (y_f - L_Sf' x_S), then x back in dof order
template <typename T, int NC>
__device__ __noinline__ void arrow_finish(int x_r) {
  typedef __attribute__((address_space(3))) T* lds_t;
  typedef __attribute__((address_space(3))) const unsigned char* lds_b;
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  const DevModel<T>& M = myo_cmodel<T>();
  const int lane = threadIdx.x, lc = lane & 15, lq = lane >> 4;
  const int nf = M.arrow_nf, nv = M.nv;
  lds_t Hp = (lds_t)s.H, xp = (lds_t)s.Mv, xin = (lds_t)LPTR(T, x_r);
  lds_b perm = (lds_b)s.hperm;
  asm volatile("" : "+v"(Hp), "+v"(xp), "+v"(xin), "+v"(perm));
  const int myrow = lane < nv ? (int)perm[lane] : 0;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (round == 1 && nf <= 4) break;
    const int f = round ? 4 : (lq < nf ? lq : nf - 1);
    const bool writer = (round ? (lq == 0) : (lq < nf)) && lc == 0;
    const int r0 = myo_arrow_row(f, 0), r1 = myo_arrow_row(f, 1), r2 = myo_arrow_row(f, 2), r3 = myo_arrow_row(f, 3);
    const int c0 = MYO_ARROW_S + 4 * f;
    const T xs = xp[lc];
    const T w0 = Hp[r0 + lc], w1 = Hp[r1 + lc], w2 = Hp[r2 + lc], w3 = Hp[r3 + lc];
    const T i0 = Hp[r0 + c0];
    const T l10 = Hp[r1 + c0], i1 = Hp[r1 + c0 + 1];
    const T l20 = Hp[r2 + c0], l21 = Hp[r2 + c0 + 1], i2 = Hp[r2 + c0 + 2];
    const T l30 = Hp[r3 + c0], l31 = Hp[r3 + c0 + 1], l32 = Hp[r3 + c0 + 2], i3 = Hp[r3 + c0 + 3];
    const T v0 = xp[c0], v1 = xp[c0 + 1], v2 = xp[c0 + 2], v3 = xp[c0 + 3];
    const T z0 = v0 - myo_row16_sum<T>(w0 * xs), z1 = v1 - myo_row16_sum<T>(w1 * xs);
    const T z2 = v2 - myo_row16_sum<T>(w2 * xs), z3 = v3 - myo_row16_sum<T>(w3 * xs);
    const T x3 = z3 * i3, x2 = (z2 - l32 * x3) * i2, x1 = (z1 - l21 * x2 - l31 * x3) * i1, x0 = (z0 - l10 * x1 - l20 * x2 - l30 * x3) * i0;
    if (writer) { xp[c0] = x0; xp[c0 + 1] = x1; xp[c0 + 2] = x2; xp[c0 + 3] = x3; }
  }
  SYNC();
  if (lane < nv) xin[lane] = xp[myrow];
  SYNC();
}
#endif
