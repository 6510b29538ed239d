// myo_lstm_step.h — one time step of G stacked one-layer LSTMs with the recurrent product ON the matrix cores and the cell
// arithmetic as its epilogue: what `torch.bmm(h, W_hh^T)` + k_lstm_cell_fwd (and k_lstm_cell_bwd + `torch.bmm(dgates, W_hh)`)
// do in two launches per step and direction (RecurrentActorCriticPolicy's lstm_actor / lstm_critic in collect_rollouts / train,
// /root/reference/src/train/trainer.py:49-71; sb3-contrib _process_sequence).  At the sizes PPO runs them
// (512-4096 rows, H = 256) those launches are latency: 6 + 5 us for 0.5 GFLOP and 6 MB.  Here a step is ONE launch per
// direction and the pre-activation / state-gradient product never leaves the chip.
//
// Work split ("K-split"): a workgroup of four waves owns an output tile — forward 64 rows x (16 units x 4 gates), backward
// 32 rows x 32 units — and each WAVE takes one quarter of the reduction dimension for the whole tile.  Both MFMA operands are
// K-contiguous in memory (rows of W_hh / W_hh^T, rows of h / dgates), so every fragment is one 16-byte load per lane straight
// into registers, no byte is loaded twice inside a workgroup, and ALL of a wave's loads are in flight before its first MFMA
// (a step is latency, not bandwidth).  The four partial tiles meet in LDS (one barrier), then wave w finishes row tile w
// (forward) / output tile w (backward) with the cell arithmetic.  The product is computed transposed, D[unit][row], so a lane
// ends up with four CONSECUTIVE units of one row: the epilogue reads and writes 8 bytes per lane and array.
//
// v_mfma_f32_16x16x32_bf16 fragment maps (cdna_hip_programming.md §3): lane l holds A[row l&15][k = 8(l>>4) + j] and
// B[k = 8(l>>4) + j][col l&15], j = 0..7; C/D: col = l&15, row = 4(l>>4) + reg.
#pragma once
#ifndef MYO_EMU

typedef __attribute__((ext_vector_type(4))) unsigned short myo_u16x4;

__device__ __forceinline__ float lstm_bf(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ myo_bf16x8 lstm_ld8(const unsigned short* p) { return *reinterpret_cast<const myo_bf16x8*>(p); }
__device__ __forceinline__ myo_u16x4 lstm_ld4(const unsigned short* p) { return *reinterpret_cast<const myo_u16x4*>(p); }
__device__ __forceinline__ void lstm_st4(unsigned short* p, float a, float b, float c, float d) {
  myo_u16x4 v;
  v.x = myo_f2bf(a); v.y = myo_f2bf(b); v.z = myo_f2bf(c); v.w = myo_f2bf(d);
  *reinterpret_cast<myo_u16x4*>(p) = v;
}

// What the forward epilogue needs besides the product, and the epilogue itself (shared by both forward kernels).
// (cp: the cell state entering the step as fp32 — from the caller's fp32 copy, c_prev32, where there is one: the state an LSTM carries
//  through an episode is the CELL state, and a bf16 round trip per step puts 2^-9 of it back in every step of a 300-step episode; h is
//  the matrix cores' operand and stays bf16)
struct LstmFwdIn { myo_u16x4 xi, xf, xg, xo; myo_f32x4 cp; float keep; };
template <int H>
__device__ __forceinline__ LstmFwdIn lstm_fwd_in(const unsigned short* __restrict__ gx, long long gx_sg, long long gx_sr,
                                                 const unsigned short* __restrict__ c_prev, const float* __restrict__ c_prev32,
                                                 const float* __restrict__ keep_next, int N, int g, int r, int u0) {
  LstmFwdIn in;
  const unsigned short* gp = gx + (size_t)g * gx_sg + (size_t)r * gx_sr + u0;
  in.xi = lstm_ld4(gp); in.xf = lstm_ld4(gp + H); in.xg = lstm_ld4(gp + 2 * H); in.xo = lstm_ld4(gp + 3 * H);
  const size_t e = ((size_t)g * N + r) * H + u0;
  if (c_prev32) in.cp = *reinterpret_cast<const myo_f32x4*>(c_prev32 + e);
  else { const myo_u16x4 c = lstm_ld4(c_prev + e); in.cp = myo_f32x4{lstm_bf(c.x), lstm_bf(c.y), lstm_bf(c.z), lstm_bf(c.w)}; }
  in.keep = keep_next ? keep_next[r] : 1.f;
  return in;
}
template <int H>
__device__ __forceinline__ void lstm_fwd_out(const LstmFwdIn& in, const myo_f32x4 (&acc)[4], int N, int g, int r, int u0,
                                             unsigned short* __restrict__ out_h, long long out_sg, unsigned short* __restrict__ hm_next,
                                             unsigned short* __restrict__ cm_next, unsigned short* __restrict__ c_new,
                                             unsigned short* __restrict__ ws, float* __restrict__ cm_next32) {
  const size_t e = ((size_t)g * N + r) * H + u0;
  const float k = in.keep;
  float iv[4], fv[4], gv[4], ov[4], cv[4], hv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    iv[j] = myo_sigmoid(lstm_bf(in.xi[j]) + acc[0][j]);
    fv[j] = myo_sigmoid(lstm_bf(in.xf[j]) + acc[1][j]);
    gv[j] = tanhf(lstm_bf(in.xg[j]) + acc[2][j]);
    ov[j] = myo_sigmoid(lstm_bf(in.xo[j]) + acc[3][j]);
    cv[j] = fv[j] * in.cp[j] + iv[j] * gv[j];
    hv[j] = ov[j] * tanhf(cv[j]);
  }
  lstm_st4(out_h + (size_t)g * out_sg + (size_t)r * H + u0, hv[0], hv[1], hv[2], hv[3]);
  lstm_st4(hm_next + e, hv[0] * k, hv[1] * k, hv[2] * k, hv[3] * k);
  if (cm_next) lstm_st4(cm_next + e, cv[0] * k, cv[1] * k, cv[2] * k, cv[3] * k);
  if (cm_next32) *reinterpret_cast<myo_f32x4*>(cm_next32 + e) = myo_f32x4{cv[0] * k, cv[1] * k, cv[2] * k, cv[3] * k};
  if (c_new) lstm_st4(c_new + e, cv[0], cv[1], cv[2], cv[3]);
  if (ws) {
    unsigned short* wp = ws + ((size_t)g * N + r) * 4 * H + u0;
    lstm_st4(wp, iv[0], iv[1], iv[2], iv[3]);
    lstm_st4(wp + H, fv[0], fv[1], fv[2], fv[3]);
    lstm_st4(wp + 2 * H, gv[0], gv[1], gv[2], gv[3]);
    lstm_st4(wp + 3 * H, ov[0], ov[1], ov[2], ov[3]);
  }
}

// forward: gates = gx + h_prev . W_hh^T -> (i, f, g, o) -> c_new = f c_prev + i g, h = o tanh(c_new).
// gx element (g, r, col) at gx[g*gx_sg + r*gx_sr + col]; out_h element (g, r, u) at out_h[g*out_sg + r*H + u]; every other
// array contiguous [G, N, .].  c_new / ws may be NULL (rollout: nothing is kept for a backward pass).
// Grid (ceil(N/64), H/16, G).  H >= 128: wave w reduces k in [w H/4, (w+1) H/4) for all 4 row tiles x 4 gate tiles.
template <int H>
__global__ void __launch_bounds__(256) k_lstm_step_fwd(const unsigned short* __restrict__ gx, long long gx_sg, long long gx_sr,
                                                       const unsigned short* __restrict__ h_prev, const unsigned short* __restrict__ c_prev,
                                                       const unsigned short* __restrict__ w_hh, const float* __restrict__ keep_next, int N,
                                                       unsigned short* __restrict__ out_h, long long out_sg,
                                                       unsigned short* __restrict__ hm_next, unsigned short* __restrict__ cm_next,
                                                       unsigned short* __restrict__ c_new, unsigned short* __restrict__ ws,
                                                       const float* __restrict__ c_prev32, float* __restrict__ cm_next32) {
  static_assert(H % 128 == 0, "K-split forward needs H / 4 to be a multiple of the MFMA's K = 32");
  constexpr int KSW = H / 128;                      // k-steps of 32 per wave
  __shared__ myo_f32x4 red[4 * 3 * 4 * 64];         // [wave][other row tile][gate q][lane]: 48 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.z, ut = blockIdx.y, row0 = blockIdx.x * 64;
  const int lr = lane & 15, lk = lane >> 4;
  const int u0 = ut * 16 + 4 * lk;
  // operands of this wave's K quarter: 4 gate tiles of W_hh rows, 4 row tiles of h_prev
  const unsigned short* W = w_hh + (size_t)g * 4 * H * H + (size_t)(ut * 16 + lr) * H + wave * (H / 4) + lk * 8;
  const unsigned short* hp = h_prev + (size_t)g * N * H + wave * (H / 4) + lk * 8;
  myo_bf16x8 a[KSW][4], b[KSW][4];
#pragma unroll
  for (int kk = 0; kk < KSW; ++kk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[kk][q] = lstm_ld8(W + (size_t)q * H * H + kk * 32);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int r = row0 + n * 16 + lr;
      b[kk][n] = lstm_ld8(hp + (size_t)(r < N ? r : N - 1) * H + kk * 32);
    }
  }
  // the epilogue's operands (row tile `wave`) travel with them
  const int r_out = row0 + wave * 16 + lr;
  const int r_ld = r_out < N ? r_out : N - 1;
  const LstmFwdIn in = lstm_fwd_in<H>(gx, gx_sg, gx_sr, c_prev, c_prev32, keep_next, N, g, r_ld, u0);
  myo_f32x4 acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[n][q] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < KSW; ++kk)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[n][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][q], b[kk][n], acc[n][q], 0, 0, 0);
  // partial tiles of the other waves' row tiles -> LDS (slot = position among the three others); own row tile stays in registers
  myo_f32x4 sum[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {                      // (n is a compile-time index: no dynamic register indexing)
    if (n == wave) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sum[q] = acc[n][q];
    } else {
      const int slot = wave - (wave > n);            // this wave's position among the three writers of row tile n
#pragma unroll
      for (int q = 0; q < 4; ++q) red[((n * 3 + slot) * 4 + q) * 64 + lane] = acc[n][q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int slot = 0; slot < 3; ++slot)
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] += red[((wave * 3 + slot) * 4 + q) * 64 + lane];
  if (r_out < N) lstm_fwd_out<H>(in, sum, N, g, r_out, u0, out_h, out_sg, hm_next, cm_next, c_new, ws, cm_next32);
}

// forward for H < 128 (the reduction is too short to split): one wave = 16 units x 16 rows, whole K.
template <int H>
__global__ void __launch_bounds__(256) k_lstm_step_fwd_small(const unsigned short* __restrict__ gx, long long gx_sg, long long gx_sr,
                                                             const unsigned short* __restrict__ h_prev, const unsigned short* __restrict__ c_prev,
                                                             const unsigned short* __restrict__ w_hh, const float* __restrict__ keep_next, int N,
                                                             unsigned short* __restrict__ out_h, long long out_sg,
                                                             unsigned short* __restrict__ hm_next, unsigned short* __restrict__ cm_next,
                                                             unsigned short* __restrict__ c_new, unsigned short* __restrict__ ws,
                                                             const float* __restrict__ c_prev32, float* __restrict__ cm_next32) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.z, ut = blockIdx.y;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  if (row0 >= N) return;
  const int lr = lane & 15, lk = lane >> 4;
  const int r_out = row0 + lr, r_ld = r_out < N ? r_out : N - 1, u0 = ut * 16 + 4 * lk;
  const unsigned short* W = w_hh + (size_t)g * 4 * H * H + (size_t)(ut * 16 + lr) * H + lk * 8;
  const unsigned short* hp = h_prev + ((size_t)g * N + r_ld) * H + lk * 8;
  const LstmFwdIn in = lstm_fwd_in<H>(gx, gx_sg, gx_sr, c_prev, c_prev32, keep_next, N, g, r_ld, u0);
  myo_f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < H / 32; ++ks) {
    const myo_bf16x8 b = lstm_ld8(hp + ks * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lstm_ld8(W + (size_t)q * H * H + ks * 32), b, acc[q], 0, 0, 0);
  }
  if (r_out < N) lstm_fwd_out<H>(in, acc, N, g, r_out, u0, out_h, out_sg, hm_next, cm_next, c_new, ws, cm_next32);
}

// backward: dh = dout + keep_next (dgates_next . W_hh), dc = keep_next dcm_next + dh o (1 - tanh^2 c_new) -> dgates, dc_prev.
// w_hh_t: bf16 [G, H, 4H] (rows of W_hh^T: the K = 4H reduction is contiguous).  dgates_next == NULL at the last step.
// dout element (g, r, u) at dout[g*dout_sg + r*H + u] (may be NULL).
// Grid (ceil(N/32), H/32, G): tile 32 rows x 32 units = 2 x 2 MFMA tiles; wave w reduces k in [w H, (w+1) H) of K = 4H for all
// four, then finishes tile (m, n) = (w >> 1, w & 1).
template <int H>
__global__ void __launch_bounds__(256) k_lstm_step_bwd(const unsigned short* __restrict__ dout, long long dout_sg,
                                                       const unsigned short* __restrict__ dg_next, const unsigned short* __restrict__ dcm_next,
                                                       const unsigned short* __restrict__ w_hh_t, const float* __restrict__ keep_next,
                                                       const unsigned short* __restrict__ c_prev, const unsigned short* __restrict__ c_new,
                                                       const unsigned short* __restrict__ ws, int N, unsigned short* __restrict__ dgates,
                                                       unsigned short* __restrict__ dc_prev) {
  constexpr int KSW = H / 32;                       // k-steps of 32 per wave (a quarter of K = 4H)
  __shared__ myo_f32x4 red[4 * 3 * 64];             // [tile][other wave][lane]: 12 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.z, ub = blockIdx.y * 32, row0 = blockIdx.x * 32;
  const int lr = lane & 15, lk = lane >> 4;
  const int tm = wave >> 1, tn = wave & 1;          // the tile this wave finishes
  const int r = row0 + tn * 16 + lr, u0 = ub + tm * 16 + 4 * lk;
  const bool live = r < N;
  const int rl = live ? r : N - 1;
  myo_bf16x8 a[KSW][2], b[KSW][2];
  if (dg_next) {
    const unsigned short* A = w_hh_t + (size_t)g * 4 * H * H + (size_t)(ub + lr) * 4 * H + wave * H + lk * 8;
    const unsigned short* B = dg_next + (size_t)g * N * 4 * H + wave * H + lk * 8;
#pragma unroll
    for (int kk = 0; kk < KSW; ++kk) {
#pragma unroll
      for (int m = 0; m < 2; ++m) a[kk][m] = lstm_ld8(A + (size_t)m * 16 * 4 * H + kk * 32);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int rr = row0 + n * 16 + lr;
        b[kk][n] = lstm_ld8(B + (size_t)(rr < N ? rr : N - 1) * 4 * H + kk * 32);
      }
    }
  }
  // the epilogue's operands travel with them
  const size_t e = ((size_t)g * N + rl) * H + u0;
  const unsigned short* wp = ws + ((size_t)g * N + rl) * 4 * H + u0;
  const myo_u16x4 wi = lstm_ld4(wp), wf = lstm_ld4(wp + H), wg = lstm_ld4(wp + 2 * H), wo = lstm_ld4(wp + 3 * H);
  const myo_u16x4 cn = lstm_ld4(c_new + e), cp = lstm_ld4(c_prev + e);
  myo_u16x4 dov = {0, 0, 0, 0}, dcv = {0, 0, 0, 0};
  if (dout) dov = lstm_ld4(dout + (size_t)g * dout_sg + (size_t)rl * H + u0);
  if (dcm_next) dcv = lstm_ld4(dcm_next + e);
  const float k = keep_next ? keep_next[rl] : 1.f;
  myo_f32x4 sum = myo_f32x4{0.f, 0.f, 0.f, 0.f};
  if (dg_next) {                                    // (uniform over the grid)
    myo_f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KSW; ++kk)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][m], b[kk][n], acc[m][n], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {                    // (t is a compile-time index: no dynamic register indexing)
      if (t == wave) sum = acc[t >> 1][t & 1];
      else red[(t * 3 + (wave - (wave > t))) * 64 + lane] = acc[t >> 1][t & 1];
    }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < 3; ++slot) sum += red[(wave * 3 + slot) * 64 + lane];
  }
  if (!live) return;
  float di[4], df[4], dg[4], dO[4], dcp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float dh = lstm_bf(dov[j]) + k * sum[j];
    const float i = lstm_bf(wi[j]), f = lstm_bf(wf[j]), gg = lstm_bf(wg[j]), o = lstm_bf(wo[j]);
    const float tc = tanhf(lstm_bf(cn[j]));
    const float dct = k * lstm_bf(dcv[j]) + dh * o * (1.f - tc * tc);
    di[j] = dct * gg * i * (1.f - i);
    df[j] = dct * lstm_bf(cp[j]) * f * (1.f - f);
    dg[j] = dct * i * (1.f - gg * gg);
    dO[j] = dh * tc * o * (1.f - o);
    dcp[j] = dct * f;
  }
  unsigned short* dp = dgates + ((size_t)g * N + r) * 4 * H + u0;
  lstm_st4(dp, di[0], di[1], di[2], di[3]);
  lstm_st4(dp + H, df[0], df[1], df[2], df[3]);
  lstm_st4(dp + 2 * H, dg[0], dg[1], dg[2], dg[3]);
  lstm_st4(dp + 3 * H, dO[0], dO[1], dO[2], dO[3]);
  lstm_st4(dc_prev + ((size_t)g * N + r) * H + u0, dcp[0], dcp[1], dcp[2], dcp[3]);
}

template <int H>
static void lstm_step_fwd_launch(const unsigned short* gx, long long gx_sg, long long gx_sr, const unsigned short* h_prev,
                                 const unsigned short* c_prev, const unsigned short* w_hh, const float* keep_next, int G, int N,
                                 unsigned short* out_h, long long out_sg, unsigned short* hm_next, unsigned short* cm_next,
                                 unsigned short* c_new, unsigned short* ws, const float* c_prev32, float* cm_next32, hipStream_t s) {
  const dim3 grid((N + 63) / 64, H / 16, G);
  if constexpr (H % 128 == 0)
    hipLaunchKernelGGL((k_lstm_step_fwd<H>), grid, dim3(256), 0, s, gx, gx_sg, gx_sr, h_prev, c_prev, w_hh, keep_next, N, out_h, out_sg, hm_next,
                       cm_next, c_new, ws, c_prev32, cm_next32);
  else
    hipLaunchKernelGGL((k_lstm_step_fwd_small<H>), grid, dim3(256), 0, s, gx, gx_sg, gx_sr, h_prev, c_prev, w_hh, keep_next, N, out_h, out_sg,
                       hm_next, cm_next, c_new, ws, c_prev32, cm_next32);
}
template <int H>
static void lstm_step_bwd_launch(const unsigned short* dout, long long dout_sg, const unsigned short* dg_next, const unsigned short* dcm_next,
                                 const unsigned short* w_hh_t, const float* keep_next, const unsigned short* c_prev,
                                 const unsigned short* c_new, const unsigned short* ws, int G, int N, unsigned short* dgates,
                                 unsigned short* dc_prev, hipStream_t s) {
  const dim3 grid((N + 31) / 32, H / 32, G);
  hipLaunchKernelGGL((k_lstm_step_bwd<H>), grid, dim3(256), 0, s, dout, dout_sg, dg_next, dcm_next, w_hh_t, keep_next, c_prev, c_new, ws, N,
                     dgates, dc_prev);
}
#endif
