// myo_lstm_seq.h — ALL time steps of G stacked one-layer LSTMs over a minibatch of sequences in ONE launch per direction: what
// the recurrent PPO minibatch step (rl/fused_lstm.py, RecurrentPPO.train of /root/reference/src/train/trainer.py:49-71 /
// sb3-contrib _process_sequence) does with T launches of myo_lstm_step_fwd and T of myo_lstm_step_bwd.  At PPO's sizes
// (512 sequences, H = 256, T = 128) such a launch is 8 us of latency — ramp-up, one dependent load -> MFMA -> reduce -> epilogue
// chain, drain — for 0.5 GFLOP; 256 of them are two thirds of the minibatch step.
//
// Work split ("row ownership"): the recurrence couples the UNITS of a row (h_t of all H units enters every gate of step t + 1) but
// never two rows.  A workgroup of eight waves therefore owns 16 rows (sequences) of one LSTM for the whole sequence and each wave
// owns H / 8 units with their four gates: no workgroup ever waits for another one, h_t (forward) / dgates_t (backward) of the 16
// rows go from one step to the next through LDS (one or two barriers per step), c_t / dc_t stay in registers.
//
// What a step then costs is the WEIGHTS: a wave multiplies its rows of W_hh (forward: 4 gates x H / 8 units, K = H) or of W_hh^T
// (backward: H / 8 units, K = 4H) with the shared 16-row tile every step — H^2 / 2 bytes per wave, 4 H^2 per workgroup and step.
// Streamed from L2 that is 9.5 us a step at H = 256 (measured, round 5: one CU pulls ~54 GB/s), slower than the per-step launches,
// which spread the same bytes over four times as many CUs.  So the weights live ON the CU, in three tiers (LstmSeqCfg, in
// fragments = the KB a wave loads for one MFMA; a wave's share is 16 fragments at H = 128, 64 at H = 256):
//   R  registers: loaded once, before the first step (H = 128: all of it, 64 registers a lane; H = 256: ~150 of a wave's 256);
//   L  LDS: copied once into the wave's own region (H = 256: 144 KB forward / 120 KB backward of the CU's 160 KB);
//   S  streamed from L2 every step, the remainder (H = 256: 8 fragments forward, 12 backward — 64 / 96 KB a step and CU), in two
//      phases through half as many registers: phase 0 is issued a whole step ahead of its use, phase 1 one product ahead.
// The weights arrive FRAGMENT-MAJOR (the caller permutes W_hh once per minibatch step, rl/fused_lstm.py lstm_seq_weights): the
// 64 x 16 bytes one wave loads for one MFMA operand are one contiguous KB in lane order; with W_hh's own row-major layout a load
// touches 16 rows x 64 bytes — 64 tag look-ups for 1 KB.  The product is computed transposed, D[unit][row] (A = weights, B = the
// tile), as in myo_lstm_step.h: a lane (lr = lane & 15, lk = lane >> 4) ends up with rows 4 lk .. 4 lk + 3 of every tile for row lr
// of the minibatch, all four gates in its own registers — the cell arithmetic needs no exchange.  WHICH units a tile's rows are is
// free (the permutation of W's rows): wave w has UT = H / 128 tiles and row i of its tile ut is unit
//   unit(w, ut, i) = 16 UT w + 4 UT (i / 4) + 4 ut + (i % 4)                      (RS = 1; the general form: below)
// so that a lane's 4 UT units are CONSECUTIVE (u0 = 16 UT w + 4 UT lk): 16-byte accesses and 64-byte row segments at H = 256.
//   forward:  w_frag[g][w][kk][q][ut][lane][j] = W_hh[g][q H + unit(w, ut, lane & 15)][32 kk + 8 (lane >> 4) + j]      (kk < H / 32)
//   backward: wt_frag[g][w][kk][ut][lane][j]   = W_hh[g][32 kk + 8 (lane >> 4) + j][unit(w, ut, lane & 15)]           (kk < 4H / 32)
// What moves besides: measured with the weights on the CU (H = 256, forward), the step's 64 KB of stores in 8-byte pieces — 16 rows x
// 32 bytes per wave instruction — cost 3.1 us of 7.6, the scattered gx loads 2.3.  So the arrays only these two kernels read (c_new,
// the gate activations ws, the masked cell state cm from slot 1 on) are TILE-MAJOR — the 4 UT values of a lane contiguous, lanes
// consecutive, then gate, wave, 16-row tile: every wave store / load is one contiguous 512 bytes or KB —
//   x_tm [t][g][rt][w][lane][4 UT]  and  ws_tm [t][g][rt][w][q][lane][4 UT]      (rt = row / 16; same sizes as [T, G, N, H] / [T, G, N, 4H])
// and the row-major ones other kernels read or write go through the LDS tiles where one exists: hm (the h tile) and dgates (the
// dgates tile) leave as whole rows, 16 bytes a lane; out_h / gx / dout are accessed in place, 4 UT units a lane.
// The cell arithmetic is what a step costs once the weights are on the CU (~110 VALU instructions a unit with ten transcendentals, on
// eight waves): a workgroup may therefore own FEWER rows, 16 / RS with RS = 1, 2 or 4 (more workgroups, more CUs).  The MFMA's 16
// columns then hold every row RS times (lane lr reads tile row lr % (16 / RS)), so RS lanes end up with the same four units of the
// same row — and lane copy s = lr / (16 / RS) takes the s-th 4 / RS of them: no cross-lane traffic, 1 / RS of the arithmetic a lane.
// With CL = 4 UT / RS cells a lane and unit(w, ut, i) = 16 UT w + 4 UT (i / 4) + CL (b / (4 / RS)) + (4 / RS) ut + b % (4 / RS), b = i % 4,
// a lane's cells are still consecutive units (u0 = 16 UT w + 4 UT lk + CL s); tile-major: x_tm[(g, row tile)][w][lane][CL].
// Roundings are those of the step kernels (state and gate activations in bf16, fp32 accumulation); the order of the fp32 sums and
// the last bits of tanh / sigmoid differ.
#pragma once
#ifndef MYO_EMU

template <int H> struct LstmSeqCfg {
  static_assert(H == 128 || H == 256, "sequence kernels: hidden size 128 or 256");
  static constexpr int NW = 8;                      // waves per workgroup
  static constexpr int UT = H / 128;                // unit tiles (of 16) per wave
  static constexpr int KF = H / 32;                 // forward k-steps (K = H)
  static constexpr int KB = 4 * H / 32;             // backward k-steps (K = 4H)
  static constexpr int FSTR = H + 8;                // LDS row stride of the h tile (bf16 elements): rows 4 banks apart
  static constexpr int BSTR = 4 * H + 8;            // ... of the dgates tile
  // tiers, in FRAGMENTS (one KB per wave each; fragment f of a wave = (kk, q, ut) forward / (kk, ut) backward, kk-major):
  // [0, 2 SB) streamed through SB fragments of registers in two phases — phase 0 = [0, SB) is consumed first (it landed during the
  // previous step) and its registers take phase 1 = [SB, 2 SB), which is consumed last, after the L and R tiers, and re-issued as
  // phase 0 of the next step; [2 SB, 2 SB + L) LDS; the rest registers.
#ifndef MYO_SEQ_FSB         /* (developer A/B: -DMYO_SEQ_FSB=.. -DMYO_SEQ_BSB=..) */
#define MYO_SEQ_FSB 4
#endif
#ifndef MYO_SEQ_BSB
#define MYO_SEQ_BSB 5
#endif
  static constexpr int FFRAG = 4 * UT, BFRAG = UT;  // fragments per k-step and wave
  static constexpr int FN = KF * FFRAG, BN = KB * BFRAG;               // ... per step and wave (H = 128: 16, H = 256: 64, either direction)
  static constexpr int FSB = H == 128 ? 0 : MYO_SEQ_FSB, FL = H == 128 ? 0 : 18, FR = FN - 2 * FSB - FL;
  static constexpr int BSB = H == 128 ? 0 : MYO_SEQ_BSB, BL = H == 128 ? 0 : 15, BR = BN - 2 * BSB - BL;
  static constexpr size_t F_LDS = ((size_t)16 * FSTR + (size_t)NW * FL * 512) * 2;      // bytes: ONE h tile, then the L tier
  static constexpr size_t B_LDS = ((size_t)16 * BSTR + (size_t)NW * BL * 512) * 2;      // ONE dgates tile, then the L tier
#ifndef MYO_SEQ_PF
#define MYO_SEQ_PF (H == 128)
#endif
  static constexpr bool PREFETCH = MYO_SEQ_PF;      // the epilogue's operands one step ahead in registers (H = 256 has none to spare)
  static_assert(F_LDS <= 160 * 1024 && B_LDS <= 160 * 1024 && FR >= 0 && BR >= 0, "tiers");
};


// two floats -> two bf16 (round to nearest even, as myo_f2bf) in one instruction: v_cvt_pk_bf16_f32 (the cell arithmetic of a step is
// ~110 VALU instructions a unit on 8 waves a CU — with ten transcendentals what a step costs once the weights are on the CU — and
// eight of its results are rounded to bf16)
typedef __attribute__((ext_vector_type(2))) __bf16 lstm_seq_bf2;
typedef __attribute__((ext_vector_type(2))) float lstm_seq_f2;
template <int H, int RS> struct LstmSeqRows {
  static_assert(RS == 1 || RS == 2 || (RS == 4 && H == 256), "row split: 1, 2, or 4 at H = 256 (a lane keeps at least two cells)");
  static constexpr int ROWS = 16 / RS;              // rows (sequences) per workgroup
  static constexpr int CL = 4 * (H / 128) / RS;     // cells (consecutive units) per lane
  static constexpr int BPU = 4 / RS;                // D rows per lane copy and unit tile
  static constexpr size_t F_LDS = ((size_t)ROWS * LstmSeqCfg<H>::FSTR + (size_t)8 * LstmSeqCfg<H>::FL * 512) * 2;
  static constexpr size_t B_LDS = ((size_t)ROWS * LstmSeqCfg<H>::BSTR + (size_t)8 * LstmSeqCfg<H>::BL * 512) * 2;
};
// D row b = BPU s + blo of a lane's four: a register index that depends on the lane copy s -> selects
template <int RS> __device__ __forceinline__ float lstm_seq_pick(const myo_f32x4& a, int s, int blo) {
  if constexpr (RS == 1) return a[blo];
  else if constexpr (RS == 2) return s ? a[2 + blo] : a[blo];
  else return (s & 2) ? ((s & 1) ? a[3] : a[2]) : ((s & 1) ? a[1] : a[0]);
}
__device__ __forceinline__ unsigned lstm_seq_pk(float a, float b) {
  const lstm_seq_bf2 v = __builtin_convertvector(lstm_seq_f2{a, b}, lstm_seq_bf2);
  return *reinterpret_cast<const unsigned*>(&v);
}
__device__ __forceinline__ float lstm_seq_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float lstm_seq_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// tanh / sigmoid without branches and without IEEE division (a branch in the time-step loop makes the compiler drain every load in
// flight at the loop head): 1 - 2 / (e^2x + 1) and 1 / (1 + e^-x) on v_exp / v_rcp, exact limits at +-inf, absolute error ~1e-7 —
// the results are stored as bf16
__device__ __forceinline__ float lstm_seq_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
__device__ __forceinline__ float lstm_seq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// The time-step loops' barrier: LDS traffic only.  __syncthreads() is a workgroup-scope fence — `s_waitcnt vmcnt(0)` in front of the
// s_barrier: every global store of the step acknowledged, twice a step, ~1 us each — and nothing in these loops hands GLOBAL data
// from one wave to another
__device__ __forceinline__ void lstm_seq_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// a lane's CL bf16 values of one array: 4, 8 or 16 bytes
template <int UT> struct LstmSeqVec { typedef unsigned short type __attribute__((ext_vector_type(UT))); };
template <int UT> __device__ __forceinline__ typename LstmSeqVec<UT>::type lstm_seq_ldv(const unsigned short* p) {
  return *reinterpret_cast<const typename LstmSeqVec<UT>::type*>(p);
}
template <int UT> __device__ __forceinline__ void lstm_seq_stv(unsigned short* p, const typename LstmSeqVec<UT>::type& v) {
  *reinterpret_cast<typename LstmSeqVec<UT>::type*>(p) = v;
}

// forward.  gx element (t, g, r, col) at gx[t gx_st + g gx_sg + r gx_sr + col]; hm [T + 1, G, N, H] row-major: slot 0 = the masked
// state entering step 0 (given), slot t + 1 written; cm [T + 1, G, N, H]: slot 0 row-major (given), slot t + 1 written TILE-MAJOR;
// keep [T, N] (0 where an episode starts at that step; step t masks with keep[t + 1], the last step with 1); out_h element
// (g, t, r, u) at out_h[g out_sg + t out_st + r H + u]; c_new [T, G, N, H], ws [T, G, N, 4H] tile-major.
// N a multiple of 16.  Grid (N / 16, G), 8 waves, LstmSeqCfg<H>::F_LDS bytes of dynamic LDS.
template <int H, int RS>
__global__ void __launch_bounds__(512) k_lstm_seq_fwd(
    const unsigned short* __restrict__ gx, long long gx_st, long long gx_sg, long long gx_sr, unsigned short* __restrict__ hm,
    unsigned short* __restrict__ cm, const unsigned short* __restrict__ w_frag, const float* __restrict__ keep, int N, int T,
    unsigned short* __restrict__ out_h, long long out_sg, long long out_st, unsigned short* __restrict__ c_new,
    unsigned short* __restrict__ ws, const float* __restrict__ c0_32) {
  typedef LstmSeqCfg<H> Cf;
  typedef LstmSeqRows<H, RS> Rw;
  constexpr int UT = Cf::UT, NF = Cf::FFRAG, NT = Cf::FN, SB = Cf::FSB, NL = Cf::FL, NR = Cf::FR, NV = Rw::CL, ROWS = Rw::ROWS, BPU = Rw::BPU;
  typedef typename LstmSeqVec<NV>::type vec;
  extern __shared__ __attribute__((aligned(16))) unsigned short lstm_seq_lds[];
  unsigned short* hbuf = lstm_seq_lds;                                  // [ROWS * FSTR]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short* wl = lstm_seq_lds + ROWS * Cf::FSTR + (size_t)wave * NL * 512 + lane * 8;      // this wave's L tier
  const int g = blockIdx.y, G = gridDim.y, row0 = blockIdx.x * ROWS;
  const int lr = (lane & 15) % ROWS, sc = (lane & 15) / ROWS, lk = lane >> 4;       // tile row, lane copy, D row quad
  const int r = row0 + lr;
  const size_t GNH = (size_t)G * N * H;
  // fragment (kk, q, ut) of this wave: the KB at ((kk 4 + q) UT + ut) 512 + lane 8 of its block
  const unsigned short* W = w_frag + (size_t)g * 4 * H * H + (size_t)wave * NT * 512 + lane * 8;
  myo_bf16x8 as[SB > 0 ? SB : 1], ar[NR];
#pragma unroll
  for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + f * 512);
#pragma unroll
  for (int f = 0; f < NL; ++f) *reinterpret_cast<myo_bf16x8*>(wl + f * 512) = lstm_ld8(W + (2 * SB + f) * 512);
#pragma unroll
  for (int f = 0; f < NR; ++f) ar[f] = lstm_ld8(W + (2 * SB + NL + f) * 512);
  // whole rows of the 16-row tile, 16 bytes a thread: hm[0] -> LDS here, the LDS tile -> hm[t + 1] after every step
  constexpr int CPR = H / 8;                                           // 16-byte chunks per row
  const int c_row = threadIdx.x / CPR, c_col = (threadIdx.x % CPR) * 8;
  const bool c_on = threadIdx.x < ROWS * CPR;
  const unsigned c_off = (unsigned)(((size_t)g * N + row0 + c_row) * H + c_col);
  if (c_on) *reinterpret_cast<myo_bf16x8*>(&hbuf[c_row * Cf::FSTR + c_col]) = lstm_ld8(hm + c_off);
  // per-lane offsets are 32-bit with a uniform 64-bit base per array and step: `global_* v, v_off, s[base]`
  const int u0 = (wave * 16 + 4 * lk) * UT + sc * NV;                  // this lane's NV consecutive units
  const unsigned oh = (unsigned)((size_t)r * H + u0), ox = (unsigned)((size_t)r * gx_sr + u0);
  // tile-major: x_tm[(g, rt)][w][lane][NV], ws_tm[(g, rt)][w][q][lane][NV]
  const unsigned ot = (unsigned)((((size_t)g * (N / ROWS) + blockIdx.x) * 8 + wave) * 64 * NV + lane * NV);
  const unsigned otw = (unsigned)((((size_t)g * (N / ROWS) + blockIdx.x) * 8 + wave) * 4 * 64 * NV + lane * NV);
  // the cell state a lane carries from step to step: float32 when the caller passes the unrounded state entering step 0 (c0_32), else
  // re-read from the bfloat16 slot every step (the step kernels' roundings on a bfloat16 c)
  const bool c_f32 = c0_32 != nullptr;                                 // (uniform)
  float cp[NV];
  if (c_f32) {
#pragma unroll
    for (int i = 0; i < NV; ++i) cp[i] = c0_32[(size_t)g * N * H + oh + i];
  } else {
    const vec cpv = lstm_seq_ldv<NV>(cm + (size_t)g * N * H + oh);    // slot 0: row-major
#pragma unroll
    for (int i = 0; i < NV; ++i) cp[i] = lstm_bf(cpv[i]);
  }
  vec xn[4];
  if constexpr (Cf::PREFETCH) {
#pragma unroll
    for (int q = 0; q < 4; ++q) xn[q] = lstm_seq_ldv<NV>(gx + (size_t)g * gx_sg + ox + q * H);
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const unsigned short* hb = hbuf + lr * Cf::FSTR + lk * 8;
    int zoff = 0;                                                  // (a loop-invariant address would have the S and L tiers' loads hoisted out of
    asm volatile("" : "+s"(zoff));                                 //  the loop — and spilled; an offset keeps the pointers' address spaces)
    // the epilogue's operands: in flight under the product
    const unsigned short* gxt = gx + (size_t)t * gx_st + (size_t)g * gx_sg;        // (uniform)
    vec x[4];
    if constexpr (Cf::PREFETCH) {                    // this step's were loaded a step ago; the next step's go out now (gx is behind HBM)
      const unsigned short* gxn = gx + (size_t)(t + 1 < T ? t + 1 : t) * gx_st + (size_t)g * gx_sg;
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[q] = xn[q]; xn[q] = lstm_seq_ldv<NV>(gxn + ox + q * H); }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = lstm_seq_ldv<NV>(gxt + ox + q * H);
    }
    const float k = t + 1 < T ? keep[(size_t)(t + 1) * N + r] : 1.f;
    myo_f32x4 acc[4][UT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) acc[q][ut] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
    // one MFMA per fragment f = (kk, q, ut); order: S phase 0, L, R, S phase 1 (tiers: LstmSeqCfg)
    auto mma = [&](int f, const myo_bf16x8& a) {
      const myo_bf16x8 b = *reinterpret_cast<const myo_bf16x8*>(hb + (f / NF) * 32);      // (the compiler keeps it across one k-step's fragments)
      acc[(f % NF) / UT][f % UT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(f % NF) / UT][f % UT], 0, 0, 0);
    };
#pragma unroll
    for (int f = 0; f < SB; ++f) mma(f, as[f]);
    if constexpr (SB > 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + zoff + (SB + f) * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < NL; ++f) mma(2 * SB + f, *reinterpret_cast<const myo_bf16x8*>(wl + zoff + f * 512));
#pragma unroll
    for (int f = 0; f < NR; ++f) mma(2 * SB + NL + f, ar[f]);
    if constexpr (SB > 0) {
#pragma unroll
      for (int f = 0; f < SB; ++f) mma(SB + f, as[f]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + zoff + f * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
    lstm_seq_barrier();                              // every wave has read h_t
    unsigned iv[NV / 2], fv[NV / 2], gv[NV / 2], ov[NV / 2], cnv[NV / 2], hv[NV / 2], hmv[NV / 2], cmv[NV / 2];      // bf16 pairs
#pragma unroll
    for (int p2 = 0; p2 < NV / 2; ++p2) {
      float vi[2], vf[2], vg[2], vo[2], c[2], h[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int i = 2 * p2 + s, ut = i / BPU, j = i % BPU;
        vi[s] = lstm_seq_sigmoid(lstm_bf(x[0][i]) + lstm_seq_pick<RS>(acc[0][ut], sc, j));
        vf[s] = lstm_seq_sigmoid(lstm_bf(x[1][i]) + lstm_seq_pick<RS>(acc[1][ut], sc, j));
        vg[s] = lstm_seq_tanh(lstm_bf(x[2][i]) + lstm_seq_pick<RS>(acc[2][ut], sc, j));
        vo[s] = lstm_seq_sigmoid(lstm_bf(x[3][i]) + lstm_seq_pick<RS>(acc[3][ut], sc, j));
        c[s] = vf[s] * cp[i] + vi[s] * vg[s];
        h[s] = vo[s] * lstm_seq_tanh(c[s]);
      }
      iv[p2] = lstm_seq_pk(vi[0], vi[1]); fv[p2] = lstm_seq_pk(vf[0], vf[1]); gv[p2] = lstm_seq_pk(vg[0], vg[1]); ov[p2] = lstm_seq_pk(vo[0], vo[1]);
      cnv[p2] = lstm_seq_pk(c[0], c[1]); hv[p2] = lstm_seq_pk(h[0], h[1]);
      hmv[p2] = lstm_seq_pk(h[0] * k, h[1] * k); cmv[p2] = lstm_seq_pk(c[0] * k, c[1] * k);
      cp[2 * p2] = c_f32 ? c[0] * k : lstm_seq_lo(cmv[p2]); cp[2 * p2 + 1] = c_f32 ? c[1] * k : lstm_seq_hi(cmv[p2]);
    }
    auto V = [](const unsigned (&w)[NV / 2]) -> const vec& { return *reinterpret_cast<const vec*>(w); };
    lstm_seq_stv<NV>(&hbuf[lr * Cf::FSTR + u0], V(hmv));
    lstm_seq_stv<NV>(out_h + (size_t)g * out_sg + (size_t)t * out_st + oh, V(hv));
    lstm_seq_stv<NV>(cm + (size_t)(t + 1) * GNH + ot, V(cmv));
    lstm_seq_stv<NV>(c_new + (size_t)t * GNH + ot, V(cnv));
    unsigned short* wst = ws + (size_t)t * 4 * GNH + otw;
    lstm_seq_stv<NV>(wst, V(iv));
    lstm_seq_stv<NV>(wst + 64 * NV, V(fv));
    lstm_seq_stv<NV>(wst + 2 * 64 * NV, V(gv));
    lstm_seq_stv<NV>(wst + 3 * 64 * NV, V(ov));
    lstm_seq_barrier();
    if (c_on) *reinterpret_cast<myo_bf16x8*>(hm + (size_t)(t + 1) * GNH + c_off) = *reinterpret_cast<const myo_bf16x8*>(&hbuf[c_row * Cf::FSTR + c_col]);
  }
}

// backward.  dout element (g, t, r, u) at dout[g dout_sg + t dout_st + r H + u]; keep / cm / c_new / ws as the forward pass left
// them (tile-major but cm's slot 0); dgates [T, G, N, 4H] row-major written.  Step t: dh = dout_t + keep[t + 1] (dgates_{t+1} . W_hh),
// dc = keep[t + 1] dc_prev_{t+1} + dh o (1 - tanh^2 c_new) -> dgates_t, dc_prev_t (the last step: no product, keep = 1).
// LstmSeqCfg<H>::B_LDS bytes of dynamic LDS (ONE dgates tile: the step's product reads it, a barrier, the step's epilogue rewrites it,
// a barrier, and it leaves for dgates[t] as whole rows).
template <int H, int RS>
__global__ void __launch_bounds__(512) k_lstm_seq_bwd(
    const unsigned short* __restrict__ dout, long long dout_sg, long long dout_st, const unsigned short* __restrict__ wt_frag,
    const float* __restrict__ keep, const unsigned short* __restrict__ cm, const unsigned short* __restrict__ c_new,
    const unsigned short* __restrict__ ws, int N, int T, unsigned short* __restrict__ dgates) {
  typedef LstmSeqCfg<H> Cf;
  typedef LstmSeqRows<H, RS> Rw;
  constexpr int UT = Cf::UT, NF = Cf::BFRAG, NT = Cf::BN, SB = Cf::BSB, NL = Cf::BL, NR = Cf::BR, NV = Rw::CL, ROWS = Rw::ROWS, BPU = Rw::BPU;
  typedef typename LstmSeqVec<NV>::type vec;
  extern __shared__ __attribute__((aligned(16))) unsigned short lstm_seq_lds[];
  unsigned short* dgbuf = lstm_seq_lds;                                 // [ROWS * BSTR]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short* wl = lstm_seq_lds + ROWS * Cf::BSTR + (size_t)wave * NL * 512 + lane * 8;
  const int g = blockIdx.y, G = gridDim.y, row0 = blockIdx.x * ROWS;
  const int lr = (lane & 15) % ROWS, sc = (lane & 15) / ROWS, lk = lane >> 4;       // tile row, lane copy, D row quad
  const int r = row0 + lr;
  const size_t GNH = (size_t)G * N * H;
  // fragment (kk, ut) of this wave: the KB at (kk UT + ut) 512 + lane 8 of its block
  const unsigned short* W = wt_frag + (size_t)g * 4 * H * H + (size_t)wave * NT * 512 + lane * 8;
  myo_bf16x8 as[SB > 0 ? SB : 1], ar[NR];
#pragma unroll
  for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + f * 512);
#pragma unroll
  for (int f = 0; f < NL; ++f) *reinterpret_cast<myo_bf16x8*>(wl + f * 512) = lstm_ld8(W + (2 * SB + f) * 512);
#pragma unroll
  for (int f = 0; f < NR; ++f) ar[f] = lstm_ld8(W + (2 * SB + NL + f) * 512);
  const int u0 = (wave * 16 + 4 * lk) * UT + sc * NV;                  // (offsets: see the forward kernel)
  const unsigned oh = (unsigned)((size_t)r * H + u0);
  const unsigned ot = (unsigned)((((size_t)g * (N / ROWS) + blockIdx.x) * 8 + wave) * 64 * NV + lane * NV);
  const unsigned otw = (unsigned)((((size_t)g * (N / ROWS) + blockIdx.x) * 8 + wave) * 4 * 64 * NV + lane * NV);
  // the dgates tile leaves as whole rows: 4H / 8 chunks of 16 bytes a row, ROWS rows, 512 threads -> ROWS H / 1024 chunks a thread
  constexpr int CPT = ROWS * H / 1024;
  float dcn[NV];                                     // dc_prev of step t + 1 (bf16-rounded, as the step kernels keep it)
#pragma unroll
  for (int i = 0; i < NV; ++i) dcn[i] = 0.f;
  const unsigned short* db = dgbuf + lr * Cf::BSTR + lk * 8;
  vec wn[4], cnn, cpn, don;
  if constexpr (Cf::PREFETCH) {
    const int tt = T - 1;
    const unsigned short* wst = ws + (size_t)tt * 4 * GNH + otw;
#pragma unroll
    for (int q = 0; q < 4; ++q) wn[q] = lstm_seq_ldv<NV>(wst + q * 64 * NV);
    cnn = lstm_seq_ldv<NV>(c_new + (size_t)tt * GNH + ot);
    cpn = tt > 0 ? lstm_seq_ldv<NV>(cm + (size_t)tt * GNH + ot) : lstm_seq_ldv<NV>(cm + (size_t)g * N * H + oh);
    don = lstm_seq_ldv<NV>(dout + (size_t)g * dout_sg + (size_t)tt * dout_st + oh);
  }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const bool last = t == T - 1;
    int zoff = 0;
    asm volatile("" : "+s"(zoff));                                 // (as in the forward kernel)
    vec wv[4], cnv, cpv, dov;
    auto load_ops = [&](int tt, vec (&w4)[4], vec& cn1, vec& cp1, vec& do1) {
      const unsigned short* wst = ws + (size_t)tt * 4 * GNH + otw;
#pragma unroll
      for (int q = 0; q < 4; ++q) w4[q] = lstm_seq_ldv<NV>(wst + q * 64 * NV);
      cn1 = lstm_seq_ldv<NV>(c_new + (size_t)tt * GNH + ot);
      cp1 = tt > 0 ? lstm_seq_ldv<NV>(cm + (size_t)tt * GNH + ot) : lstm_seq_ldv<NV>(cm + (size_t)g * N * H + oh);      // (slot 0: row-major)
      do1 = lstm_seq_ldv<NV>(dout + (size_t)g * dout_sg + (size_t)tt * dout_st + oh);
    };
    if constexpr (Cf::PREFETCH) {                    // (as in the forward kernel)
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = wn[q];
      cnv = cnn; cpv = cpn; dov = don;
      load_ops(t > 0 ? t - 1 : 0, wn, cnn, cpn, don);
    } else load_ops(t, wv, cnv, cpv, dov);
    const float k = last ? 1.f : keep[(size_t)(t + 1) * N + r];
    myo_f32x4 acc[UT];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) acc[ut] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {                                     // (uniform)
      auto mma = [&](int f, const myo_bf16x8& a) {
        const myo_bf16x8 b = *reinterpret_cast<const myo_bf16x8*>(db + (f / NF) * 32);
        acc[f % UT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[f % UT], 0, 0, 0);
      };
#pragma unroll
      for (int f = 0; f < SB; ++f) mma(f, as[f]);
      if constexpr (SB > 0) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + zoff + (SB + f) * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int f = 0; f < NL; ++f) mma(2 * SB + f, *reinterpret_cast<const myo_bf16x8*>(wl + zoff + f * 512));
#pragma unroll
      for (int f = 0; f < NR; ++f) mma(2 * SB + NL + f, ar[f]);
      if constexpr (SB > 0) {
#pragma unroll
        for (int f = 0; f < SB; ++f) mma(SB + f, as[f]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < SB; ++f) as[f] = lstm_ld8(W + zoff + f * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lstm_seq_barrier();                              // every wave has read dgates_{t+1}
    unsigned d_i[NV / 2], d_f[NV / 2], d_g[NV / 2], d_o[NV / 2];      // bf16 pairs
#pragma unroll
    for (int p2 = 0; p2 < NV / 2; ++p2) {
      float di[2], df[2], dg[2], dO[2], dcp[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int i = 2 * p2 + s;
        const float dh = lstm_bf(dov[i]) + k * lstm_seq_pick<RS>(acc[i / BPU], sc, i % BPU);
        const float vi = lstm_bf(wv[0][i]), vf = lstm_bf(wv[1][i]), vg = lstm_bf(wv[2][i]), vo = lstm_bf(wv[3][i]);
        const float tc = lstm_seq_tanh(lstm_bf(cnv[i]));
        const float dct = k * dcn[i] + dh * vo * (1.f - tc * tc);
        di[s] = dct * vg * vi * (1.f - vi);
        df[s] = dct * lstm_bf(cpv[i]) * vf * (1.f - vf);
        dg[s] = dct * vi * (1.f - vg * vg);
        dO[s] = dh * tc * vo * (1.f - vo);
        dcp[s] = dct * vf;
      }
      d_i[p2] = lstm_seq_pk(di[0], di[1]); d_f[p2] = lstm_seq_pk(df[0], df[1]); d_g[p2] = lstm_seq_pk(dg[0], dg[1]); d_o[p2] = lstm_seq_pk(dO[0], dO[1]);
      const unsigned dc2 = lstm_seq_pk(dcp[0], dcp[1]);
      dcn[2 * p2] = lstm_seq_lo(dc2); dcn[2 * p2 + 1] = lstm_seq_hi(dc2);
    }
    auto V = [](const unsigned (&w)[NV / 2]) -> const vec& { return *reinterpret_cast<const vec*>(w); };
    unsigned short* dl = dgbuf + lr * Cf::BSTR + u0;
    lstm_seq_stv<NV>(dl, V(d_i)); lstm_seq_stv<NV>(dl + H, V(d_f)); lstm_seq_stv<NV>(dl + 2 * H, V(d_g)); lstm_seq_stv<NV>(dl + 3 * H, V(d_o));
    lstm_seq_barrier();
    unsigned short* dgt = dgates + (size_t)t * 4 * GNH + ((size_t)g * N + row0) * 4 * H;      // (uniform: the tile's rows are contiguous)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = c * 512 + threadIdx.x, rr = ch / (H / 2), cc = (ch % (H / 2)) * 8;       // chunk ch of 16 x 4H / 8
      *reinterpret_cast<myo_bf16x8*>(dgt + (unsigned)(rr * 4 * H + cc)) = *reinterpret_cast<const myo_bf16x8*>(&dgbuf[rr * Cf::BSTR + cc]);
    }
  }
}

template <typename K>
static int lstm_seq_lds_attr(K kernel, size_t lds, bool* done) {     // above the 64 KB a kernel gets without asking
  if (*done || lds <= 48 * 1024) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  *done = true;
  return 0;
}
template <int H, int RS>
static int lstm_seq_fwd_launch(const unsigned short* gx, long long gx_st, long long gx_sg, long long gx_sr, unsigned short* hm,
                               unsigned short* cm, const unsigned short* w_frag, const float* keep, int G, int N, int T, unsigned short* out_h,
                               long long out_sg, long long out_st, unsigned short* c_new, unsigned short* ws, const float* c0_32, hipStream_t s) {
  static bool attr = false;
  typedef LstmSeqRows<H, RS> Rw;
  if (lstm_seq_lds_attr(&k_lstm_seq_fwd<H, RS>, Rw::F_LDS, &attr)) return 1;
  hipLaunchKernelGGL((k_lstm_seq_fwd<H, RS>), dim3(N / Rw::ROWS, G), dim3(512), Rw::F_LDS, s, gx, gx_st, gx_sg, gx_sr, hm, cm, w_frag, keep, N, T,
                     out_h, out_sg, out_st, c_new, ws, c0_32);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int H, int RS>
static int lstm_seq_bwd_launch(const unsigned short* dout, long long dout_sg, long long dout_st, const unsigned short* wt_frag,
                               const float* keep, const unsigned short* cm, const unsigned short* c_new, const unsigned short* ws, int G,
                               int N, int T, unsigned short* dgates, hipStream_t s) {
  static bool attr = false;
  typedef LstmSeqRows<H, RS> Rw;
  if (lstm_seq_lds_attr(&k_lstm_seq_bwd<H, RS>, Rw::B_LDS, &attr)) return 1;
  hipLaunchKernelGGL((k_lstm_seq_bwd<H, RS>), dim3(N / Rw::ROWS, G), dim3(512), Rw::B_LDS, s, dout, dout_sg, dout_st, wt_frag, keep, cm, c_new, ws,
                     N, T, dgates);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#endif
