// myo_ppo_mlp.h — one PPO minibatch step of the MLP actor-critic (obs -> H -> H -> head, ReLU; both nets) on the
// matrix cores: what SB3's PPO.train does per minibatch through autograd (evaluate_actions, clipped surrogate + value
// loss, backward; /root/reference/src/train/trainer.py:57-71 -> sb3 PPO.train) as FOUR launches
// instead of ~25 (hipBLASLt GEMMs + elementwise kernels, each 8-20 us for microseconds of work):
//
//   k_mlp_prep     bf16 operand images of the weights in the layouts the MFMA fragments want (padded, plus transposes)
//   k_mlp_fwdbwd   per 64-row block and net: gather -> layer 1 -> layer 2 -> head -> PPO loss gradient -> d hidden 2 ->
//                  d hidden 1; activations stay in LDS between layers; every activation / activation-gradient leaves
//                  ONCE, feature-major ([feature][row], bf16), which is the K-contiguous layout the weight-gradient
//                  GEMMs read straight into MFMA fragments
//   k_mlp_wgrad    dW = dY' X for the six weight matrices and the four hidden bias vectors: 128 x 128 output tiles,
//                  split over the batch (split-K) into per-split slabs — plain stores, no float atomics
//   k_mlp_reduce   flat gradient = sum of the slabs in split order (deterministic)
// followed by the existing k_colmajor_finish (loss sums, log_std and head-bias gradients) and myo_adam_clip_step.
//
// v_mfma_f32_16x16x32_bf16 fragment maps (cdna_hip_programming.md §3): lane l holds A[row l&15][k = 8(l>>4) + j] and
// B[k = 8(l>>4) + j][col l&15], j = 0..7; C/D: col = l&15, row = 4(l>>4) + reg.  Both operands are therefore read as
// 16 contiguous bytes per lane from a K-contiguous image (rows of X / H in LDS, rows of W in global memory).
#pragma once
#ifndef MYO_EMU

typedef __attribute__((ext_vector_type(8))) __bf16 myo_bf16x8;
typedef __attribute__((ext_vector_type(4))) float myo_f32x4;

#define MLP_H 256          // hidden width of both trunks
#define MLP_OPMAX 128      // padded observation width (multiple of 32)
#define MLP_APM 48         // head outputs padded to 3 tiles of 16
#define MLP_AKP 64         // head width as the K of the d-hidden GEMM
#define MLP_XS (MLP_OPMAX + 8)
#define MLP_HS (MLP_H + 8)
#define MLP_DS (MLP_AKP + 8)
#define MLP_SOS 49
#define MLP_SPLITK 16

struct MlpArgs {
  const float *obs, *act, *oldlp, *adv, *ret;      // rollout arrays [N, .]
  const long long* idx;                            // minibatch rows [B]
  const float *log_std, *adv_stats;
  const unsigned short *W1p, *W2, *W2T, *Whp, *WhT;  // bf16 images: [2][H][OP], [2][H][H], [2][H][H], [2][APM][H], [2][H][AKP]
  const float* bias;                               // fp32 [2][H] b1, [2][H] b2, [2][APM] bh
  unsigned short *XT, *H1T, *H2T, *dH1T, *dH2T, *dOT;   // feature-major bf16 [rows][B]: 128, 2H, 2H, 2H, 2H, 2*128
  float* part;                                     // loss partials, column-major [(2A+3)][NB]
  int B, O, A, OP, NB;
  float clip, vf_coef;
};

__device__ __forceinline__ unsigned short mlp_f2bf(float x) {   // round-to-nearest-even (finite inputs)
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float mlp_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct MlpPrepArgs {
  const float* p;                                  // flat fp32 parameters
  long long off_W1[2], off_b1[2], off_W2[2], off_b2[2], off_Wh[2], off_bh[2];
  int O, OP, Ah[2];                                // head widths: A (actor), 1 (critic)
  unsigned short *W1p, *W2, *W2T, *Whp, *WhT;
  float* bias;
  const float* adv_part;       // [3 * adv_nb] slices (n, mean, M2) of the advantage moments left by k_adv_moments, or adv_nb = 0
  float* adv_stats;
  int adv_nb, B;
};

#define MLP_ADV_BLOCKS 64
__global__ void __launch_bounds__(256) k_mlp_prep(MlpPrepArgs P) {
  // the last block first merges the slices of the advantage moments in slice order (Chan's update, a fixed order: deterministic)
  // — here, behind a kernel boundary, instead of by the last block of k_adv_moments behind a device-scope fence (an L2 write-back:
  // 12.8 us for the launch) — while the other blocks convert the weights
  if (P.adv_nb > 0 && blockIdx.x == gridDim.x - 1) {
    __shared__ float sp[3 * MLP_ADV_BLOCKS];
    const int t = threadIdx.x;
    if (t < 3 * P.adv_nb) sp[t] = P.adv_part[t];
    __syncthreads();
    if (t == 0) {
      float cn = 0.f, cm = 0.f, c2 = 0.f;
      for (int b = 0; b < P.adv_nb; ++b) {
        const float bn = sp[3 * b], bm = sp[3 * b + 1], b2 = sp[3 * b + 2];
        if (bn > 0.f) {
          const float tot = cn + bn, dl = bm - cm;
          cm += dl * (bn / tot);
          c2 += b2 + dl * dl * (cn * bn / tot);
          cn = tot;
        }
      }
      P.adv_stats[0] = cm; P.adv_stats[1] = sqrtf(c2 / (P.B > 1 ? P.B - 1 : 1));
    }
  }
  constexpr int H = MLP_H;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  for (int net = 0; net < 2; ++net) {
    const float* W1 = P.p + P.off_W1[net]; const float* W2 = P.p + P.off_W2[net]; const float* Wh = P.p + P.off_Wh[net];
    for (long long e = tid; e < (long long)H * P.OP; e += nth) {
      const int o = (int)(e / P.OP), i = (int)(e % P.OP);
      P.W1p[(size_t)net * H * P.OP + e] = i < P.O ? mlp_f2bf(W1[(size_t)o * P.O + i]) : (unsigned short)0;
    }
    for (long long e = tid; e < (long long)H * H; e += nth) {
      const int o = (int)(e / H), i = (int)(e % H);
      const unsigned short v = mlp_f2bf(W2[e]);
      P.W2[(size_t)net * H * H + e] = v;
      P.W2T[(size_t)net * H * H + (size_t)i * H + o] = v;
    }
    for (long long e = tid; e < (long long)MLP_APM * H; e += nth) {
      const int a = (int)(e / H), h = (int)(e % H);
      P.Whp[(size_t)net * MLP_APM * H + e] = a < P.Ah[net] ? mlp_f2bf(Wh[(size_t)a * H + h]) : (unsigned short)0;
    }
    for (long long e = tid; e < (long long)H * MLP_AKP; e += nth) {
      const int h = (int)(e / MLP_AKP), a = (int)(e % MLP_AKP);
      P.WhT[(size_t)net * H * MLP_AKP + e] = a < P.Ah[net] ? mlp_f2bf(Wh[(size_t)a * H + h]) : (unsigned short)0;
    }
    for (long long e = tid; e < H; e += nth) {
      P.bias[net * H + e] = P.p[P.off_b1[net] + e];
      P.bias[2 * H + net * H + e] = P.p[P.off_b2[net] + e];
    }
    for (long long e = tid; e < MLP_APM; e += nth) P.bias[4 * H + net * MLP_APM + e] = e < P.Ah[net] ? P.p[P.off_bh[net] + e] : 0.f;
  }
}

// mean and unbiased std of adv[idx[0..B)), first half: MLP_ADV_BLOCKS blocks (a minibatch drawn from a shuffled rollout touches one
// 128 B line per row — 2 MB through ONE CU's L1 was 23 us of a 150 us optimizer step), each reducing its contiguous slice of the
// index list to (n, mean, M2) with the values kept in registers (one memory pass) -> part[3 * block]; k_mlp_prep merges them.
__global__ void __launch_bounds__(256) k_adv_moments(const float* __restrict__ adv, const long long* __restrict__ idx, int B,
                                                     float* __restrict__ part) {
  __shared__ float red[4];
  __shared__ float s_mean;
  const int t = threadIdx.x, nb = gridDim.x;
  const int chunk = (B + nb - 1) / nb, lo = blockIdx.x * chunk, hi = min(B, lo + chunk), n = max(hi - lo, 0);
  constexpr int PER = 4;                  // slices up to 1024 rows in registers; longer ones re-read (second loops below)
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) { const int i = lo + t + 256 * k; v[k] = i < hi ? adv[idx[i]] : 0.f; s += v[k]; }
  for (int i = lo + t + 256 * PER; i < hi; i += 256) s += adv[idx[i]];
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) s_mean = n ? ((red[0] + red[1]) + (red[2] + red[3])) / n : 0.f;
  __syncthreads();
  const float mean = s_mean;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) { const float d = v[k] - mean; m2 += (lo + t + 256 * k) < hi ? d * d : 0.f; }
  for (int i = lo + t + 256 * PER; i < hi; i += 256) { const float d = adv[idx[i]] - mean; m2 += d * d; }
  for (int off = 32; off >= 1; off >>= 1) m2 += __shfl_xor(m2, off, 64);
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = m2;
  __syncthreads();
  if (t == 0) { part[3 * blockIdx.x] = (float)n; part[3 * blockIdx.x + 1] = mean; part[3 * blockIdx.x + 2] = (red[0] + red[1]) + (red[2] + red[3]); }
}

// One wave's slab of a layer: BM rows x 64 columns, acc[mt][nt] += A(LDS rows, K-contiguous) * W(global rows, K-contiguous)'.
// The weight fragments of a WHOLE slab (KSTEPS x 4 x 16 B per lane) are requested by mlp_load_w one phase ahead of the MFMAs
// that consume them (mlp_mma_slab): a layer's weights were written by the previous launch on whichever XCD ran k_mlp_prep, so
// the first workgroups of every other XCD miss their L2 on each line; requested next to their use that was up to eight serial
// misses per layer (8.8 us of a 33 us workgroup), requested a phase ahead they fly during the previous epilogue and barrier.
template <int KSTEPS>
__device__ __forceinline__ void mlp_load_w(myo_bf16x8 (&bw)[8][4], const unsigned short* Wg, int w_stride, int n0, int lm, int lq) {
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bw[ks][nt] = *reinterpret_cast<const myo_bf16x8*>(Wg + (size_t)(n0 + 16 * nt + lm) * w_stride + 32 * ks + 8 * lq);
  __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the loads back to their uses to save registers)
}
template <int KSTEPS, int MT>
__device__ __forceinline__ void mlp_mma_slab(myo_f32x4 (&acc)[MT][4], const unsigned short* As, int as_stride, const myo_bf16x8 (&bw)[8][4],
                                             int lm, int lq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    // (two row tiles' A fragments at a time: with all four in registers next to 64 accumulators and the 128 weight registers
    // of a K = 256 slab the kernel spilled)
#pragma unroll
    for (int m2 = 0; m2 < MT; m2 += 2) {
      myo_bf16x8 a[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const myo_bf16x8*>(As + (16 * (m2 + mt) + lm) * as_stride + 32 * ks + 8 * lq);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m2 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bw[ks][nt], acc[m2 + mt][nt], 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef MLP_PROF
__device__ unsigned long long g_mlp_prof[16];
#define MLP_STAMP(k) { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 3 && blockIdx.y == 0) g_mlp_prof[k] = wall_clock64(); }
extern "C" int myo_debug_mlp_prof(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_prof), sizeof g_mlp_prof); }
#else
#define MLP_STAMP(k)
#endif

// LDS of one workgroup (BM rows): row ids, X (later the head output), ONE hidden image (H1, then H2, then dH2 — each is only
// needed as the A operand of the next GEMM; the ReLU masks the two gradient epilogues need stay in registers, one bit per
// accumulator element of the lane, because the wave that applies a mask is the wave that produced the activation),
// actions (later d log_std terms), d(head output), log_std / exp(-log_std), adv / old log-prob / return.
// BM = 32: 38 KB, 241 VGPRs -> two workgroups per CU (the register budget decides), whose phases (gather, MFMA slabs, the
// one-wave loss phase, feature-major stores) overlap each other's latencies.  BM = 64 (75 KB, also two per CU, every weight
// fragment fetched from L2 feeding four row tiles instead of two) was measured at the SAME time per row — 54 us per 64-row
// workgroup against 2 x 28 us, k_mlp_fwdbwd 71.0 vs 69.8 us at B = 16384 — with 33 spilled registers: the phases scale with
// the rows (epilogue conversions, 2-byte LDS stores, strided feature-major stores), not with the weight traffic.
#ifndef MLP_BM
#define MLP_BM 32
#endif
#define MLP_FWDBWD_LDS (512 + MLP_BM * MLP_XS * 2 + MLP_BM * MLP_HS * 2 + MLP_BM * MLP_SOS * 4 + MLP_BM * MLP_DS * 2 + 128 * 4 + 192 * 4)

// activation epilogue of one slab: bias + ReLU -> bf16 -> row-major LDS image (next layer's A operand) + feature-major global copy
// returns the lane's ReLU mask: bit 16 mt + 4 nt + r = (activation != 0)
template <int MT>
__device__ __forceinline__ unsigned long long mlp_store_act(const myo_f32x4 (&acc)[MT][4], const float (&bb)[4], unsigned short* Hs, unsigned short* HT,
                                                            size_t B, int r0, int n0, int lm, int lq) {
  static_assert(MT * 16 <= 64, "one mask bit per accumulator element");
  unsigned long long mask = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + 16 * nt + lm, m = 16 * mt + 4 * lq;
      unsigned short h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h[r] = mlp_f2bf(fmaxf(acc[mt][nt][r] + bb[nt], 0.f)); Hs[(m + r) * MLP_HS + n] = h[r];
        mask |= (unsigned long long)(h[r] != 0) << (16 * mt + 4 * nt + r);
      }
      *reinterpret_cast<uint2*>(HT + (size_t)n * B + r0 + m) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    }
  return mask;
}
// gradient epilogue: * (H > 0) without a branch (the mask is all-ones / zero bits) -> bf16 -> optional LDS image + feature-major global copy
template <int MT, bool TO_LDS>
__device__ __forceinline__ void mlp_store_grad(const myo_f32x4 (&acc)[MT][4], unsigned long long mask, unsigned short* Hs, unsigned short* HT, size_t B,
                                               int r0, int n0, int lm, int lq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + 16 * nt + lm, m = 16 * mt + 4 * lq;
      unsigned short h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned short on = (unsigned short)(-(int)((mask >> (16 * mt + 4 * nt + r)) & 1ull));
        h[r] = mlp_f2bf(acc[mt][nt][r]) & on;
        if (TO_LDS) Hs[(m + r) * MLP_HS + n] = h[r];
      }
      *reinterpret_cast<uint2*>(HT + (size_t)n * B + r0 + m) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    }
}

__global__ void __launch_bounds__(256, 2) k_mlp_fwdbwd(MlpArgs P) {
  constexpr int H = MLP_H, XS = MLP_XS, HS = MLP_HS, DS = MLP_DS, SOS = MLP_SOS, BM = MLP_BM, MT = BM / 16, RW = BM / 4;
  extern __shared__ __align__(16) unsigned char mlp_smem[];
  long long* s_idx = reinterpret_cast<long long*>(mlp_smem);                      // [BM] (64 slots)
  unsigned short* Xs = reinterpret_cast<unsigned short*>(mlp_smem + 512);         // [BM][XS] bf16; later So fp32 [BM][SOS]
  float* So = reinterpret_cast<float*>(mlp_smem + 512);
  unsigned short* Hs = Xs + BM * XS;                                              // [BM][HS]: H1, then H2, then dH2
  float* Sa = reinterpret_cast<float*>(Hs + BM * HS);                             // [BM][SOS] actions, then dlogp (z^2 - 1) in place
  unsigned short* dOs = reinterpret_cast<unsigned short*>(Sa + BM * SOS);         // [BM][DS]
  float* s_ls = reinterpret_cast<float*>(dOs + BM * DS);                          // [64] log_std, [64] exp(-log_std)
  float* s_row = s_ls + 128;                                                      // [3][64] adv / old log-prob / return of the block's rows
  static_assert(BM * MLP_SOS * 4 <= BM * MLP_XS * 2, "the head output tile fits where X was");
  static_assert(BM == 32 || BM == 64, "wave / tile mapping below");
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, lm = lane & 15, lq = lane >> 4;
  const int net = blockIdx.y, blk = blockIdx.x, r0 = blk * BM, B = P.B, O = P.O, A = P.A, OP = P.OP;
  const int n0 = 64 * w;
  MLP_STAMP(0)
  if (t < BM) s_idx[t] = P.idx[r0 + t];
  if (t < A) { const float ls = P.log_std[t]; s_ls[t] = ls; s_ls[64 + t] = __expf(-ls); }
  const float* b1 = P.bias + net * H; const float* b2 = P.bias + 2 * H + net * H; const float* bh = P.bias + 4 * H + net * MLP_APM;
  float bb1[4], bb2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { bb1[nt] = b1[n0 + 16 * nt + lm]; bb2[nt] = b2[n0 + 16 * nt + lm]; }
  myo_bf16x8 bw[8][4];                   // the weight fragments of the NEXT MFMA slab (see mlp_load_w)
  {
    const unsigned short* W1 = P.W1p + (size_t)net * H * OP;
    if (OP == 96) mlp_load_w<3>(bw, W1, OP, n0, lm, lq);
    else if (OP == 128) mlp_load_w<4>(bw, W1, OP, n0, lm, lq);
    else if (OP == 64) mlp_load_w<2>(bw, W1, OP, n0, lm, lq);
    else mlp_load_w<1>(bw, W1, OP, n0, lm, lq);
  }
  __syncthreads();
  if (t < BM) {
    const size_t src = (size_t)s_idx[t];
    s_row[t] = P.adv[src]; s_row[64 + t] = P.oldlp[src]; s_row[128 + t] = P.ret[src];
  }
  // ---- gather.  Wave w takes rows RW w .. RW w + RW - 1; a row is read by the whole wave (columns lane and lane + 64: coalesced),
  // and all loads of a lane are in flight before the first is used.  X -> bf16 LDS image (zero-padded to OP columns).
  {
    float v[RW][2], av[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const size_t row = (size_t)s_idx[RW * w + k];
      v[k][0] = lane < O ? P.obs[row * O + lane] : 0.f;
      v[k][1] = lane + 64 < O ? P.obs[row * O + lane + 64] : 0.f;
      av[k] = (net == 0 && lane < A) ? P.act[row * A + lane] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      Xs[(RW * w + k) * XS + lane] = mlp_f2bf(v[k][0]);
      if (lane + 64 < OP) Xs[(RW * w + k) * XS + lane + 64] = mlp_f2bf(v[k][1]);
      if (lane < A) Sa[(RW * w + k) * SOS + lane] = av[k];
    }
    if (net == 0) {   // feature-major copy of X for the weight-gradient GEMM (the critic's block has the same rows): RW rows per column and lane
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        if (c < OP) {
          unsigned u[RW / 2];
#pragma unroll
          for (int k = 0; k < RW / 2; ++k) u[k] = (unsigned)mlp_f2bf(v[2 * k][h]) | ((unsigned)mlp_f2bf(v[2 * k + 1][h]) << 16);
          uint4* dst = reinterpret_cast<uint4*>(P.XT + (size_t)c * B + r0 + RW * w);
#pragma unroll
          for (int q = 0; q < RW / 8; ++q) dst[q] = make_uint4(u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]);
        }
      }
    }
  }
  __syncthreads();
  MLP_STAMP(1)
  myo_f32x4 acc[MT][4];
  // ---- layer 1 (K = OP, a multiple of 32 up to 128)
  if (OP == 96) mlp_mma_slab<3, MT>(acc, Xs, XS, bw, lm, lq);
  else if (OP == 128) mlp_mma_slab<4, MT>(acc, Xs, XS, bw, lm, lq);
  else if (OP == 64) mlp_mma_slab<2, MT>(acc, Xs, XS, bw, lm, lq);
  else mlp_mma_slab<1, MT>(acc, Xs, XS, bw, lm, lq);
  mlp_load_w<H / 32>(bw, P.W2 + (size_t)net * H * H, H, n0, lm, lq);
  const unsigned long long mask1 = mlp_store_act<MT>(acc, bb1, Hs, P.H1T + (size_t)net * H * B, B, r0, n0, lm, lq);
  __syncthreads();
  MLP_STAMP(2)
  // ---- layer 2
  mlp_mma_slab<H / 32, MT>(acc, Hs, HS, bw, lm, lq);
  // the head's tiles (16 rows x 16 outputs, MT x 3 of them): wave w < 3 takes output tile w of every row tile (one set of
  // weight fragments, 32 VGPRs, reused MT times)
  const unsigned short* Wh = P.Whp + (size_t)net * MLP_APM * H;
  myo_bf16x8 bh2[H / 32];
  if (w < 3) {
#pragma unroll
    for (int ks = 0; ks < H / 32; ++ks) bh2[ks] = *reinterpret_cast<const myo_bf16x8*>(Wh + (size_t)(16 * w + lm) * H + 32 * ks + 8 * lq);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                       // every wave has read H1 out of the hidden image: H2 takes its place
  const unsigned long long mask2 = mlp_store_act<MT>(acc, bb2, Hs, P.H2T + (size_t)net * H * B, B, r0, n0, lm, lq);
  __syncthreads();
  MLP_STAMP(3)
  // ---- head (fp32 into So, which reuses X's storage)
  if (w < 3) {
    const float bv = bh[16 * w + lm];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      myo_f32x4 ah = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) {
        const myo_bf16x8 a = *reinterpret_cast<const myo_bf16x8*>(Hs + (16 * mt + lm) * HS + 32 * ks + 8 * lq);
        ah = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh2[ks], ah, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) So[(16 * mt + 4 * lq + r) * SOS + 16 * w + lm] = ah[r] + bv;
    }
  }
  mlp_load_w<MLP_AKP / 32>(bw, P.WhT + (size_t)net * H * MLP_AKP, MLP_AKP, n0, lm, lq);      // d hidden 2's weights fly during the loss phase
  __syncthreads();
  MLP_STAMP(4)
  // ---- loss gradient w.r.t. the head output (k_ppo_loss's arithmetic), one row per lane of wave 0
  const int NB = P.NB;
  if (w == 0) {
    const int r = lane < BM ? lane : 0;
    const bool on = lane < BM;
    if (net == 0) {
      float logp = 0.f;
      for (int a = 0; a < A; ++a) {
        const float z = (Sa[r * SOS + a] - So[r * SOS + a]) * s_ls[64 + a];
        logp += -0.5f * z * z - s_ls[a] - 0.9189385332046727f;
      }
      const float an = (s_row[r] - P.adv_stats[0]) / (P.adv_stats[1] + 1e-8f);
      const float ratio = __expf(logp - s_row[64 + r]);
      const float s1 = an * ratio;
      const float rc = fminf(fmaxf(ratio, 1.f - P.clip), 1.f + P.clip);
      const float s2 = an * rc;
      float pl_i = on ? -fminf(s1, s2) / B : 0.f;
      const bool inside = (ratio > 1.f - P.clip) && (ratio < 1.f + P.clip);
      const float dlogp = -(an * ratio) * ((s1 <= s2) ? 1.f : (inside ? 1.f : 0.f)) / B;
      if (on) {
        for (int a = 0; a < A; ++a) {
          const float inv = s_ls[64 + a];
          const float z = (Sa[r * SOS + a] - So[r * SOS + a]) * inv;
          const float dm = dlogp * z * inv;
          So[r * SOS + a] = dm;
          Sa[r * SOS + a] = dlogp * (z * z - 1.f);
          dOs[r * DS + a] = mlp_f2bf(dm);
        }
        for (int a = A; a < MLP_AKP; ++a) dOs[r * DS + a] = 0;
      }
      for (int off = 32; off >= 1; off >>= 1) pl_i += __shfl_xor(pl_i, off, 64);
      if (lane == 0) P.part[(size_t)A * NB + blk] = pl_i;
    } else {
      const float dv = So[r * SOS] - s_row[128 + r];
      float vl_i = on ? dv * dv / B : 0.f, dv_i = on ? P.vf_coef * 2.f / B * dv : 0.f;
      if (on) {
        dOs[r * DS] = mlp_f2bf(dv_i);
        for (int a = 1; a < MLP_AKP; ++a) dOs[r * DS + a] = 0;
      }
      for (int off = 32; off >= 1; off >>= 1) { vl_i += __shfl_xor(vl_i, off, 64); dv_i += __shfl_xor(dv_i, off, 64); }
      if (lane == 0) { P.part[(size_t)(A + 1) * NB + blk] = vl_i; P.part[(size_t)(2 * A + 2) * NB + blk] = dv_i; }
    }
  }
  __syncthreads();
  MLP_STAMP(5)
  if (net == 0 && t < A) {       // column sums over the block's rows: d log_std and the action head's bias gradient
    float my_ls = 0.f, my_db = 0.f;
    for (int r = 0; r < BM; ++r) { my_ls += Sa[r * SOS + t]; my_db += So[r * SOS + t]; }
    P.part[(size_t)t * NB + blk] = my_ls; P.part[(size_t)(A + 2 + t) * NB + blk] = my_db;
  }
  {   // feature-major copy of d(head output) for the head's weight gradient
    const int Ah = net == 0 ? A : 1;
    for (int e = t; e < Ah * (BM / 8); e += 256) {
      const int c = e / (BM / 8), ch = e % (BM / 8);
      unsigned v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (unsigned)dOs[(8 * ch + 2 * k) * DS + c] | ((unsigned)dOs[(8 * ch + 2 * k + 1) * DS + c] << 16);
      *reinterpret_cast<uint4*>(P.dOT + ((size_t)net * 128 + c) * B + r0 + 8 * ch) = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
  MLP_STAMP(6)
  // ---- d hidden 2 = (dOut Wh) * (H2 > 0), written over H2's LDS image (its A-operand role ended with the head)
  mlp_mma_slab<MLP_AKP / 32, MT>(acc, dOs, DS, bw, lm, lq);
  mlp_load_w<H / 32>(bw, P.W2T + (size_t)net * H * H, H, n0, lm, lq);
  mlp_store_grad<MT, true>(acc, mask2, Hs, P.dH2T + (size_t)net * H * B, B, r0, n0, lm, lq);
  __syncthreads();
  MLP_STAMP(7)
  // ---- d hidden 1 = (dH2 W2) * (H1 > 0)
  mlp_mma_slab<H / 32, MT>(acc, Hs, HS, bw, lm, lq);
  mlp_store_grad<MT, false>(acc, mask1, Hs, P.dH1T + (size_t)net * H * B, B, r0, n0, lm, lq);
  MLP_STAMP(8)
}

// ---- rollout inference + action sampling in one launch (collect_rollouts' policy call: SB3 ActorCriticPolicy.forward ->
// DiagGaussianDistribution.sample / log_prob, RolloutBuffer.add): the forward half of k_mlp_fwdbwd on the rows of the policy
// input itself (no index gather), then, for the actor's blocks, the Philox draws of k_sample_actions (same counters: env, group
// of four actions, draw) on the fp32 head output — the arithmetic the optimizer step's forward repeats, so the first epoch's
// probability ratio is exp(0) up to the bf16 weights having moved.  The critic's blocks write the value.  Replaces, per env
// step, myo_rollout_policy_input + two batched trunk GEMMs + two bias/ReLU kernels + two head GEMMs + myo_rollout_sample.
struct MlpPolicyArgs {
  const float* obs;                                 // [N][O] policy input (normalised observation)
  const float* log_std;
  const unsigned short *W1p, *W2, *Whp;             // bf16 images of k_mlp_prep
  const float* bias;
  float* obs_buf;                                   // [T][N][O] rollout buffer (row t_idx written) or null
  float *act_buf, *val_buf, *logp_buf, *clipped;    // [T][N][A], [T][N], [T][N], [N][A]
  const int* t_idx;
  unsigned long long* draw_counter;                 // [2]: [0] = draw of this step, [1] <- [0] + 1 (committed by k_rollout_advance)
  unsigned long long seed;
  int N, O, A, OP, deterministic;
};
#define MLP_POLICY_LDS (MLP_BM * MLP_XS * 2 + MLP_BM * MLP_HS * 2 + 128 * 4)

// forward epilogue: bias + ReLU -> bf16 -> row-major LDS image (the next layer's A operand)
template <int MT>
__device__ __forceinline__ void mlp_store_act_lds(const myo_f32x4 (&acc)[MT][4], const float (&bb)[4], unsigned short* Hs, int n0, int lm, int lq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + 16 * nt + lm, m = 16 * mt + 4 * lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) Hs[(m + r) * MLP_HS + n] = mlp_f2bf(fmaxf(acc[mt][nt][r] + bb[nt], 0.f));
    }
}

__global__ void __launch_bounds__(256, 2) k_mlp_policy(MlpPolicyArgs P) {
  constexpr int H = MLP_H, XS = MLP_XS, HS = MLP_HS, SOS = MLP_SOS, BM = MLP_BM, MT = BM / 16, RW = BM / 4;
  extern __shared__ __align__(16) unsigned char mlp_smem[];
  unsigned short* Xs = reinterpret_cast<unsigned short*>(mlp_smem);               // [BM][XS] bf16; later So fp32 [BM][SOS]
  float* So = reinterpret_cast<float*>(mlp_smem);
  unsigned short* Hs = Xs + BM * XS;                                              // [BM][HS]: H1, then H2
  float* s_ls = reinterpret_cast<float*>(Hs + BM * HS);                           // [64] log_std, [64] exp(log_std)
  static_assert(BM * MLP_SOS * 4 <= BM * MLP_XS * 2, "the head output tile fits where X was");
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, lm = lane & 15, lq = lane >> 4;
  const int net = blockIdx.y, r0 = blockIdx.x * BM, N = P.N, O = P.O, A = P.A, OP = P.OP;
  const int n0 = 64 * w;
  const size_t tt = (size_t)(*P.t_idx);
  if (t < A) { const float ls = P.log_std[t]; s_ls[t] = ls; s_ls[64 + t] = __expf(ls); }
  const float* b1 = P.bias + net * H; const float* b2 = P.bias + 2 * H + net * H; const float* bh = P.bias + 4 * H + net * MLP_APM;
  float bb1[4], bb2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { bb1[nt] = b1[n0 + 16 * nt + lm]; bb2[nt] = b2[n0 + 16 * nt + lm]; }
  myo_bf16x8 bw[8][4];
  {
    const unsigned short* W1 = P.W1p + (size_t)net * H * OP;
    if (OP == 96) mlp_load_w<3>(bw, W1, OP, n0, lm, lq);
    else if (OP == 128) mlp_load_w<4>(bw, W1, OP, n0, lm, lq);
    else if (OP == 64) mlp_load_w<2>(bw, W1, OP, n0, lm, lq);
    else mlp_load_w<1>(bw, W1, OP, n0, lm, lq);
  }
  {   // rows of the policy input -> bf16 LDS image (zero-padded to OP columns); the actor's blocks also fill the rollout buffer
    float v[RW][2];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const size_t row = (size_t)(r0 + RW * w + k);
      v[k][0] = lane < O ? P.obs[row * O + lane] : 0.f;
      v[k][1] = lane + 64 < O ? P.obs[row * O + lane + 64] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      Xs[(RW * w + k) * XS + lane] = mlp_f2bf(v[k][0]);
      if (lane + 64 < OP) Xs[(RW * w + k) * XS + lane + 64] = mlp_f2bf(v[k][1]);
      if (net == 0 && P.obs_buf) {
        float* dst = P.obs_buf + (tt * N + (size_t)(r0 + RW * w + k)) * O;
        if (lane < O) dst[lane] = v[k][0];
        if (lane + 64 < O) dst[lane + 64] = v[k][1];
      }
    }
  }
  __syncthreads();
  myo_f32x4 acc[MT][4];
  if (OP == 96) mlp_mma_slab<3, MT>(acc, Xs, XS, bw, lm, lq);
  else if (OP == 128) mlp_mma_slab<4, MT>(acc, Xs, XS, bw, lm, lq);
  else if (OP == 64) mlp_mma_slab<2, MT>(acc, Xs, XS, bw, lm, lq);
  else mlp_mma_slab<1, MT>(acc, Xs, XS, bw, lm, lq);
  mlp_load_w<H / 32>(bw, P.W2 + (size_t)net * H * H, H, n0, lm, lq);
  mlp_store_act_lds<MT>(acc, bb1, Hs, n0, lm, lq);
  __syncthreads();
  mlp_mma_slab<H / 32, MT>(acc, Hs, HS, bw, lm, lq);
  const unsigned short* Wh = P.Whp + (size_t)net * MLP_APM * H;
  myo_bf16x8 bh2[H / 32];
  if (w < 3) {
#pragma unroll
    for (int ks = 0; ks < H / 32; ++ks) bh2[ks] = *reinterpret_cast<const myo_bf16x8*>(Wh + (size_t)(16 * w + lm) * H + 32 * ks + 8 * lq);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                       // every wave has read H1 out of the hidden image: H2 takes its place
  mlp_store_act_lds<MT>(acc, bb2, Hs, n0, lm, lq);
  __syncthreads();
  if (w < 3) {
    const float bv = bh[16 * w + lm];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      myo_f32x4 ah = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) {
        const myo_bf16x8 a = *reinterpret_cast<const myo_bf16x8*>(Hs + (16 * mt + lm) * HS + 32 * ks + 8 * lq);
        ah = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh2[ks], ah, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) So[(16 * mt + 4 * lq + r) * SOS + 16 * w + lm] = ah[r] + bv;
    }
  }
  __syncthreads();
  if (net == 1) {
    if (t < BM) P.val_buf[tt * N + r0 + t] = So[t * SOS];
    return;
  }
  // ---- the actor's blocks: 8 lanes per row, lane g = the four actions 4g .. 4g+3 (and 4(g+8) ..): one Philox block each
  const unsigned long long ctr = P.draw_counter[0];
  const int row = t >> 3, g0 = t & 7, i = r0 + row;
  float logp = 0.f;
  static_assert(BM * 8 == 256, "8 lanes per row");
  for (int a0 = 4 * g0; a0 < A; a0 += 32) {
    unsigned int c[4] = {(unsigned int)i, (unsigned int)(a0 >> 2), (unsigned int)ctr, (unsigned int)(ctr >> 32)};
    unsigned int k0 = (unsigned int)P.seed, k1 = (unsigned int)(P.seed >> 32);
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    float z[4];
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.0f * __logf(u1));
      float sn, cs;
      __sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = a0 + k;
      if (a < A) {
        const float mu = So[row * SOS + a], ls = s_ls[a];
        const float act = P.deterministic ? mu : mu + s_ls[64 + a] * z[k];
        const float zz = (act - mu) * __expf(-ls);
        logp += -0.5f * zz * zz - ls - 0.9189385332046727f;
        P.act_buf[(tt * N + i) * A + a] = act;
        P.clipped[(size_t)i * A + a] = fminf(fmaxf(act, -1.f), 1.f);
      }
    }
  }
  for (int off = 4; off >= 1; off >>= 1) logp += __shfl_xor(logp, off, 8);
  if (g0 == 0) P.logp_buf[tt * N + i] = logp;
  if (blockIdx.x == 0 && t == 0) P.draw_counter[1] = ctr + 1;
}

// ---- weight gradients.  C[m][n] = sum_r AT[m][r] BT[n][r] over the split's rows; 128 x 128 tile per workgroup (4 waves as
// 2 x 2 of 64 x 64), both operands K(row)-contiguous in global memory.  Feature-major buffers are allocated in whole 128-row
// groups, so every fragment load is in bounds; stores are masked by the true M, N.
struct MlpWgradJob { const unsigned short *AT, *BT; int M, N, ldo, m0, n0; long long out_off, bias_off; };
struct MlpWgradArgs { MlpWgradJob job[16]; int njobs, B, rows_per_split; long long G; float* slab; };

__global__ void __launch_bounds__(256) k_mlp_wgrad(MlpWgradArgs P) {
  const int jid = blockIdx.x % P.njobs, split = blockIdx.x / P.njobs;
  const MlpWgradJob J = P.job[jid];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, lm = lane & 15, lq = lane >> 4;
  const int wm = w >> 1, wn = w & 1, B = P.B;
  const int mb = J.m0 + 64 * wm, nb = J.n0 + 64 * wn;
  const size_t k0 = (size_t)split * P.rows_per_split;
  myo_f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    accb[mt] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = myo_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = J.bias_off >= 0 && nb == 0;
  myo_bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  const int ksteps = P.rows_per_split / 32;       // even (rows_per_split is a multiple of 64)
  const unsigned short* ap[4]; const unsigned short* bp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ap[q] = J.AT + (size_t)(mb + 16 * q + lm) * B + k0 + 8 * lq; bp[q] = J.BT + (size_t)(nb + 16 * q + lm) * B + k0 + 8 * lq; }
  myo_bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a0[q] = *reinterpret_cast<const myo_bf16x8*>(ap[q]); b0[q] = *reinterpret_cast<const myo_bf16x8*>(bp[q]); }
  for (int ks = 0; ks < ksteps; ks += 2) {        // two register sets: the loads of one k-step fly during the MFMAs of the other
#pragma unroll
    for (int q = 0; q < 4; ++q) { a1[q] = *reinterpret_cast<const myo_bf16x8*>(ap[q] + 32 * (ks + 1)); b1[q] = *reinterpret_cast<const myo_bf16x8*>(bp[q] + 32 * (ks + 1)); }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mt], b0[nt], acc[mt][nt], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mt], ones, accb[mt], 0, 0, 0);
    }
    if (ks + 2 < ksteps) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { a0[q] = *reinterpret_cast<const myo_bf16x8*>(ap[q] + 32 * (ks + 2)); b0[q] = *reinterpret_cast<const myo_bf16x8*>(bp[q] + 32 * (ks + 2)); }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b1[nt], acc[mt][nt], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], ones, accb[mt], 0, 0, 0);
    }
  }
  float* out = P.slab + (size_t)split * P.G;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mb + 16 * mt + 4 * lq + r, n = nb + 16 * nt + lm;
        if (m < J.M && n < J.N) out[J.out_off + (size_t)m * J.ldo + n] = acc[mt][nt][r];
      }
  if (do_bias && lm == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mb + 16 * mt + 4 * lq + r;
        if (m < J.M) out[J.bias_off + m] = accb[mt][r];
      }
  }
}

// The tail of the optimizer step's gradient in ONE launch (single-rank path: nothing is exchanged between the gradient and
// the clip): blocks 0 .. MLP_RF_BLOCKS-1 sum the split-K slabs into the flat gradient (split order) and their share of
// |g|^2; block MLP_RF_BLOCKS + c finishes column c of the loss partials (k_colmajor_finish's arithmetic: loss sums, d log_std,
// head-bias gradients) and the squares of what it wrote; sq_part[0 .. MLP_RF_BLOCKS + ncol) is what myo_adam_apply adds up in
// a fixed order; block 0 also advances Adam's step counter (k_grad_sqnorm's other job).  Instead of k_mlp_reduce +
// k_colmajor_finish + k_grad_sqnorm.
#define MLP_RF_BLOCKS 256
struct MlpRfArgs {
  const float* slab; float* g; long long G; int splits;
  const float* part; float* acc; int NB, A; float ent_coef;
  long long off_log_std, off_bh0, off_bh1;
  float* sq_part; int* adam_step;
};
__global__ void __launch_bounds__(256) k_mlp_reduce_finish(MlpRfArgs P) {
  __shared__ float red[4];
  const int t = threadIdx.x, A = P.A;
  if ((int)blockIdx.x < MLP_RF_BLOCKS) {
    float sq = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + t; e < P.G; e += (long long)MLP_RF_BLOCKS * 256) {
      float s = 0.f;
#pragma unroll 4
      for (int k = 0; k < P.splits; ++k) s += P.slab[(size_t)k * P.G + e];
      const bool theirs = (e >= P.off_log_std && e < P.off_log_std + A) || (e >= P.off_bh0 && e < P.off_bh0 + A) || e == P.off_bh1;
      if (!theirs) { P.g[e] = s; sq += s * s; }          // (the column blocks own the log_std / head-bias slots)
    }
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if ((t & 63) == 0) red[t >> 6] = sq;
    __syncthreads();
    if (t == 0) {
      P.sq_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
      if (blockIdx.x == 0) { const int done = P.adam_step[1]; P.adam_step[0] = done; P.adam_step[1] = done + 1; }
    }
    return;
  }
  const int c = (int)blockIdx.x - MLP_RF_BLOCKS;
  if (t >= 64) return;
  float a = 0.f;
  for (int b = t; b < P.NB; b += 64) a += P.part[(size_t)c * P.NB + b];
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if (t == 0) {
    P.acc[c] = a;
    float w = 0.f;
    if (c < A) { w = a - P.ent_coef; P.g[P.off_log_std + c] = w; }          // d(-ent_coef * entropy)/d log_std = -ent_coef
    else if (c >= A + 2 && c < 2 * A + 2) { w = a; P.g[P.off_bh0 + c - A - 2] = w; }
    else if (c == 2 * A + 2) { w = a; P.g[P.off_bh1] = w; }
    P.sq_part[MLP_RF_BLOCKS + c] = w * w;
  }
}

// flat gradient = sum over splits of the slabs, in split order
__global__ void __launch_bounds__(256) k_mlp_reduce(const float* __restrict__ slab, float* __restrict__ g, long long G, int splits) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= G) return;
  float s = 0.f;
#pragma unroll 4
  for (int k = 0; k < splits; ++k) s += slab[(size_t)k * G + e];
  g[e] = s;
}
#endif
