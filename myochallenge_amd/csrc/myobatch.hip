// myobatch.hip — libmyobatch (C ABI in include/myobatch.h): the HIP backend, the gfx950 kernels and every entry point.
//   hipcc --offload-arch=gfx950  -> libmyobatch.so   (the product; no CPU path)
// One workgroup = one wavefront = one environment; grid = number of environments.
// The host side it shares with the test tooling (model tables, batch records, myo_batch_create) is csrc/myo_host.h; the lane-serial
// emulation of the same kernel SOURCE is another translation unit (csrc/myobatch_emu.cpp + csrc/emu_host.h), not a path of this one.
#include <hip/hip_runtime.h>
#include <utility>
#include <vector>
#define MYO_BACKEND_NAME "gfx950"
#define MYO_BACKEND_DENSE_NEWTON 0
#define MYO_BACKEND_BATCH_FIELDS hipEvent_t ev0, ev1; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
#include "myo_host.h"

// ------------------------------------------------------------------------------------------ backend
typedef hipStream_t be_stream;
static int be_malloc(void** p, size_t n) { return (int)hipMalloc(p, n ? n : 1); }
static void be_free(void* p) { (void)hipFree(p); }
static int be_h2d(void* d, const void* h, size_t n) { return (int)hipMemcpy(d, h, n, hipMemcpyHostToDevice); }
static int be_set_device(int dev) { return (int)hipSetDevice(dev); }
static const char* be_errstr(int e) { return hipGetErrorString((hipError_t)e); }

// Model / task parameters live in __constant__ memory (scalar loads in every phase function).  __constant__
// symbols exist once PER DEVICE, so what is bound is tracked per device; they are (re)uploaded on the launch
// stream whenever a different batch launches on that device.  If the batch bound before launched on another
// stream, the upload first waits for that stream's work (its kernels may still be reading the constants).
#define MYO_MAX_DEVICES 64
struct BoundDev { const myo_batch* b = nullptr; hipStream_t last = nullptr; bool have_last = false; hipEvent_t ev = nullptr; };
static BoundDev g_bound[MYO_MAX_DEVICES];
static std::mutex g_bound_mu;
// The workspace blocks (TaskDev::ctrl_ws, myo_physics.h): MYO_WS_SLOTS x MYO_ENVWS_N doubles (~60 MB) and their owner flags
// (myo_ws_acquire / myo_ws_release, wave.h) per DEVICE, shared by every fp64 batch on it — a workgroup owns a block while it holds an
// env, so kernels of different batches, even on different streams, never meet in one.  Allocated with the first such batch, freed
// with the last.
struct SlotWs { double* p = nullptr; int* map = nullptr; int users = 0; char* big = nullptr; int big_users = 0; };
static SlotWs g_slot_ws[MYO_MAX_DEVICES];
static bool slot_workspace_acquire(int device, double** ws, int** map) {
  if (device < 0 || device >= MYO_MAX_DEVICES) return false;
  std::lock_guard<std::mutex> lk(g_bound_mu);
  SlotWs& w = g_slot_ws[device];
  if (!w.p) {
    void *q = nullptr, *mp = nullptr;
    const size_t bytes = sizeof(double) * (size_t)MYO_WS_SLOTS * MYO_ENVWS_N, mbytes = sizeof(int) * (size_t)MYO_WS_SLOTS;      // (the blocks' owner flags: 0 = free)
    if (hipMalloc(&q, bytes) != hipSuccess) return false;
    if (hipMalloc(&mp, mbytes) != hipSuccess) { (void)hipFree(q); return false; }
    if (hipMemset(q, 0, bytes) != hipSuccess || hipMemset(mp, 0, mbytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipFree(q); (void)hipFree(mp);
      return false;
    }
    w.p = (double*)q; w.map = (int*)mp; w.users = 0;
  }
  w.users++;
  *ws = w.p; *map = w.map;
  return true;
}
static void slot_workspace_release(int device) {
  if (device < 0 || device >= MYO_MAX_DEVICES) return;
  std::lock_guard<std::mutex> lk(g_bound_mu);
  SlotWs& w = g_slot_ws[device];
  if (w.users > 0 && --w.users == 0) { (void)hipFree(w.p); (void)hipFree(w.map); w.p = nullptr; w.map = nullptr; }
}
// ... and the blocks of the 48-slot fp64 scratch (TaskDev::big_ws: contact records + wrap results, MYO_BIGWS_BYTES per dense slot index,
// ~200 MB): allocated with the first batch that needs them (a die, extended collision pairs, condim 4 / 6), freed with the last
static char* big_workspace_acquire(int device) {
  if (device < 0 || device >= MYO_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_bound_mu);
  SlotWs& w = g_slot_ws[device];
  if (!w.big) {
    void* q = nullptr;
    const size_t bytes = (size_t)MYO_WS_SLOTS * MYO_BIGWS_BYTES;
    if (hipMalloc(&q, bytes) != hipSuccess) return nullptr;
    if (hipMemset(q, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(q); return nullptr; }
    w.big = (char*)q; w.big_users = 0;
  }
  w.big_users++;
  return w.big;
}
static void big_workspace_release(int device) {
  if (device < 0 || device >= MYO_MAX_DEVICES) return;
  std::lock_guard<std::mutex> lk(g_bound_mu);
  SlotWs& w = g_slot_ws[device];
  if (w.big_users > 0 && --w.big_users == 0) { (void)hipFree(w.big); w.big = nullptr; }
}
// every launch entry point runs on the batch's own device, whatever the caller's current device is
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess); }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
static void unbind(const myo_batch* b, int device) {
  std::lock_guard<std::mutex> lk(g_bound_mu);
  if (device >= 0 && device < MYO_MAX_DEVICES && g_bound[device].b == b) g_bound[device].b = nullptr;
}
static int bind_constants(myo_batch* b, hipStream_t st) {
  if (b->device < 0 || b->device >= MYO_MAX_DEVICES) return fail(MYO_E_ARG, "device index out of range");
  std::lock_guard<std::mutex> lk(g_bound_mu);
  BoundDev& g = g_bound[b->device];
  if (g.b == b) { g.last = st; g.have_last = true; return 0; }
  hipError_t e = hipSuccess;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  if (g.b && g.have_last && g.last != st && cap == hipStreamCaptureStatusNone) {
    if (!g.ev) e = hipEventCreateWithFlags(&g.ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(g.ev, g.last);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, g.ev, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipSuccess; }    // the other stream may be gone: nothing left to wait for
  }
  if (b->dtype == MYO_F64) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_model_d), &b->Md, sizeof b->Md, 0, hipMemcpyHostToDevice, st);
  else e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_model_f), &b->Mf, sizeof b->Mf, 0, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_task), &b->K, sizeof b->K, 0, hipMemcpyHostToDevice, st);
#ifdef MYO_STAGECOUNT
  { static int rep; const char* r = getenv("MYO_DBG_REPEAT"); rep = r ? atoi(r) : 0; if (e == hipSuccess) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_dbg_repeat), &rep, sizeof rep, 0, hipMemcpyHostToDevice, st); }
#endif
  if (e != hipSuccess) return fail(MYO_E_DEVICE, "binding model constants failed: %s", hipGetErrorString(e));
  g.b = b; g.last = st; g.have_last = true;
  return 0;
}

// ------------------------------------------------------------------------------------------ kernels
#define MYO_LDS_ALIGN(n) (((n) + 15) / 16 * 16)
#ifdef MYO_PROF
__device__ unsigned long long g_prof[MYO_NPROF];
extern "C" int myo_debug_read_prof(double* out16, int reset) {     /* out16: MYO_NPROF (32) doubles */
  unsigned long long h[MYO_NPROF];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof h) != hipSuccess) return -1;
  for (int k = 0; k < MYO_NPROF; ++k) out16[k] = (double)h[k];
  if (reset) { memset(h, 0, sizeof h); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof h); }
  return 0;
}
#endif
// RK4 stage storage of this workgroup: behind the scratch in LDS for the mixed stepper with the base contact capacity (18,384 +
// 1,360 B: still eight workgroups per CU; the per-workgroup block in global memory cost a global round trip in each of the
// ~5 bookkeeping phases of a stage), in global memory otherwise (fp64: 31.6 KB + 1.8 KB would lose the fifth workgroup per CU)
#define MYO_RK_IN_LDS(T, NC) (sizeof(T) == 4 && (NC) == MYO_NCON_MAX)
static_assert(MYO_LDS_ALIGN(sizeof(Scratch<float, MYO_NCON_MAX>)) + sizeof(RkScratch<float>) <= 20480, "RK4 stage storage next to the scratch: eight workgroups per CU");
template <typename T, bool RK, int NC>
__device__ __forceinline__ RkScratch<T>* rk_storage() {
  if constexpr (!RK) return nullptr;
  else if constexpr (MYO_RK_IN_LDS(T, NC)) return reinterpret_cast<RkScratch<T>*>(myo_lds + MYO_LDS_ALIGN(sizeof(Scratch<T, NC>)));
  else return reinterpret_cast<RkScratch<T>*>(c_task.rk_ws) + blockIdx.x;
}
#ifdef MYO_WGTIME
// developer diagnostic (tools/dev/gpu_wgtime.py): start / end of every workgroup of the last k_step launch on the 100 MHz wall clock
__device__ unsigned long long g_wg_time[2 * 16384];
extern "C" int myo_debug_read_wgtime(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_time), sizeof(unsigned long long) * 2 * (size_t)(n < 16384 ? n : 16384)) == hipSuccess ? 0 : -1;
}
#endif
// ---- launch order of the env steps.  A step of 4096 envs is two rounds over the chip's 2048 one-wave slots, and an env step
// takes 0.8 .. 1.6 x the mean (contacts, Newton iterations): in index order the second job of a slot starts whenever its first
// ends, the slot sums spread, and the launch waits for the worst one — measured (tools/dev/gpu_wgtime.py): 28 % of slots x
// makespan idle, the last quarter of the launch running on a fraction of the chip.  The dispatcher hands workgroups to freed
// slots in blockIdx order, so the ORDER is the only lever a one-env-per-workgroup kernel has: longest predicted first — the long
// jobs start at once, the short ones fill the slots as they free up (list scheduling, LPT).  Prediction = exponentially smoothed
// duration of the env's own last steps (k_step times every env on the 100 MHz wall clock; the step-to-step correlation of the
// durations is ~0.5, the smoothing keeps the persistent half).  The results of a step do not depend on the order.
// k_step_order: ONE block; cost <- (cost + ticks) / 2, 256 cost buckets between the batch minimum and maximum, counting sort
// (descending) -> order.  Which env comes first inside a bucket is left to the LDS atomics: irrelevant to the results.
__global__ void __launch_bounds__(1024) k_step_order(const unsigned int* __restrict__ ticks, float* __restrict__ cost, int n, int* __restrict__ order,
                                                     int* __restrict__ step_gen) {
  __shared__ float s_lo[16], s_hi[16];
  __shared__ unsigned int s_cnt[256], s_off[256];
  const int t = threadIdx.x;
  if (step_gen && t == 0) step_gen[0] = (int)((unsigned)step_gen[0] + 1u);          // the next launch is the next generation of the part protocol
  if (!order) return;
  float lo = 3.4e38f, hi = 0.f;
  for (int i = t; i < n; i += 1024) {
    const float tk = (float)ticks[i], c0 = cost[i];
    const float c = c0 > 0.f ? 0.5f * (c0 + tk) : tk;
    cost[i] = c;
    lo = fminf(lo, c); hi = fmaxf(hi, c);
  }
  for (int off = 32; off >= 1; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
  if ((t & 63) == 0) { s_lo[t >> 6] = lo; s_hi[t >> 6] = hi; }
  if (t < 256) s_cnt[t] = 0;
  __syncthreads();
  lo = s_lo[0]; hi = s_hi[0];
  for (int k = 1; k < 16; ++k) { lo = fminf(lo, s_lo[k]); hi = fmaxf(hi, s_hi[k]); }
  const float scale = hi > lo ? 255.99f / (hi - lo) : 0.f;
  for (int i = t; i < n; i += 1024) atomicAdd(&s_cnt[255 - (int)((cost[i] - lo) * scale)], 1u);
  __syncthreads();
  if (t == 0) { unsigned a = 0; for (int k = 0; k < 256; ++k) { s_off[k] = a; a += s_cnt[k]; } }
  __syncthreads();
  for (int i = t; i < n; i += 1024) order[atomicAdd(&s_off[255 - (int)((cost[i] - lo) * scale)], 1u)] = i;
}
// ---- env steps in parts.  Even in the best order a slot's two jobs add up their spreads and the launch ends with one job-long
// tail on a draining chip.  The tail is as long as the LAST job of a slot: so an env step is cut into P jobs of decreasing
// length (default 7 + 3 substeps), blocks p n .. (p+1) n - 1 running part p, handed over through the env record in HBM.  In
// dispatch order every part p is placed before any part p + 1, the long jobs run with the chip full, and what drains at the
// end is the short jobs' tail (simulated with the measured duration spread: makespan 2.50 -> 2.20 ms for 7 + 3; measured
// 2.82 -> 2.45 ms, 2.39 with the launch order above; the states are bit-identical).
// Protocol (placement- and dispatch-order-independent; MI355X_MICROARCH.md "inter-workgroup visibility").  part_state[env] =
// 16 g + 2 q: parts < q of step g are published; + 1: part q is claimed and running.  A workgroup of part p polls (relaxed,
// agent scope): a state past "part p claimed" -> nothing left to do, exit; 16 g + 2 q with q <= p -> claim q (agent-scope CAS)
// and run parts q .. p back to back (q < p only if an earlier part's block has not been placed yet: that block then finds
// its part taken and exits); 16 g + 2 q + 1 with q < p -> the producer is RUNNING somewhere: sleep and poll again — no
// workgroup ever waits for one that is not running.  Taking over a record another workgroup published: ONE agent acquire fence
// after the poll, then plain loads.  Publishing: store the record, drain the stores, agent release fence, drain, relaxed
// agent-scope store of the new state (16 (g + 1) after the last part: the next launch's generation).
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_step(EnvRecordLayout L, double* rec, const float* act,
                                             float* obs, float* rew, unsigned char* done, unsigned char* trunc,
                                             float* term_obs, float* comps, float* ep_info, unsigned char* bad_state,
                                             const int* __restrict__ order, unsigned int* __restrict__ ticks,
                                             int* part_state, const int* __restrict__ step_gen, StepPlan plan) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int nparts = part_state ? plan.nparts : 1;
  const int nenv = (int)gridDim.x / nparts;
  const int p = (int)blockIdx.x / nenv;
  const int slot = (int)blockIdx.x - p * nenv;
  const int env = order ? order[slot] : slot;
  const unsigned long long t_start = ticks ? wall_clock64() : 0ull;
#ifdef MYO_WGTIME
  if (threadIdx.x == 0 && blockIdx.x < 16384) g_wg_time[2 * blockIdx.x] = wall_clock64();
#endif
#ifdef MYO_PROF
  if (threadIdx.x == 0) { for (int k = 0; k < MYO_NPROF; ++k) s.prof[k] = 0; s.prof_t = clock64(); }
  __syncthreads();
#endif
  int q = 0;                                   // first part this workgroup runs
  int* st = nullptr;
  unsigned base = 0;
  int last = 1;
  if (part_state) {
    base = 16u * (unsigned)step_gen[0];     // (all state arithmetic is modulo 2^32: the generation counter may wrap)
    st = part_state + env;
    int from = -1;                             // decided by lane 0: the part to start from, -1 = nothing to do
    if (threadIdx.x == 0) {
      for (;;) {
        const int v = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned j = (unsigned)v - base;
        if (j > 2u * (unsigned)p) {             // part p is claimed or done: exit.  A state of ANOTHER generation (neither a part of this
          // step nor its final 16 (g + 1)) means step_gen and part_state have come apart — a failed launch, two streams stepping one
          // batch: the step would silently write nothing, so it is counted (myo_batch_protocol_errors) and the env flagged
          if (j > 2u * (unsigned)nparts && j != 16u) { if (K.health) atomicAdd(K.health, 1); if (bad_state) bad_state[env] = 1; }
          break;
        }
        if ((j & 1) == 0) {
          int expect = v;
          if (__hip_atomic_compare_exchange_strong(st, &expect, (int)((unsigned)v + 1u), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { from = (int)(j >> 1); break; }
          continue;
        }
        __builtin_amdgcn_s_sleep(32);           // an earlier part is running
      }
    }
    from = __builtin_amdgcn_readfirstlane(from);
    if (from < 0) return;
    q = from;
    if (q > 0) {                               // the record was published by another workgroup of this launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    last = p == nparts - 1;
  }
  // ONE call site — the kernel holds one copy of the step: substeps k_lo .. k_hi of the frame_skip (a launch without a plan: the whole step)
  const int k_lo = part_state ? plan.k[q] : 0, k_hi = (part_state && !last) ? plan.k[p + 1] : -1, pub = (part_state && plan.wt && !last) ? 1 : 0;
  env_step<T, RK ? 1 : 0>(M, K, L, rec + (size_t)env * L.stride, s, env, act, obs, rew, done, trunc, term_obs, comps, ep_info, bad_state, k_lo, k_hi, pub);
  if (part_state) {
    // (the duration is added BEFORE the part is published: the next part's workgroup may run, and add its own, the moment it is)
    if (ticks && threadIdx.x == 0) {
      const unsigned int d = (unsigned int)(wall_clock64() - t_start) + (q > 0 ? ticks[env] : 0u);
      if (plan.wt) __hip_atomic_store(ticks + env, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else ticks[env] = d;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!last && !plan.wt) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_store(st, (int)(last ? base + 16u : base + 2u * (unsigned)p + 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (ticks && threadIdx.x == 0) ticks[env] = (unsigned int)(wall_clock64() - t_start);
  }
#ifdef MYO_WGTIME
  if (threadIdx.x == 0 && blockIdx.x < 16384) g_wg_time[2 * blockIdx.x + 1] = wall_clock64();
#endif
#ifdef MYO_PROF
  PROF(s, 0)
  __syncthreads();
  if (threadIdx.x < MYO_NPROF) atomicAdd(&g_prof[threadIdx.x], s.prof[threadIdx.x]);
#endif
}
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_reset(EnvRecordLayout L, double* rec,
                                              const unsigned char* mask, float* obs) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int env = blockIdx.x;
  env_reset<T>(M, K, L, rec + (size_t)env * L.stride, s, env, mask, obs);
}
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_step_inner(EnvRecordLayout L, double* rec, const unsigned char* mask, const float* act,
                                                   float* obs, unsigned char* done) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int env = blockIdx.x;
  env_step_inner<T, RK ? 1 : 0>(M, K, L, rec + (size_t)env * L.stride, s, env, mask, act, obs, done);
}
// the same for a LIST of envs: block r steps env idx[r] (idx[r] < 0: an empty slot) with row r of act / obs / done
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_step_inner_idx(EnvRecordLayout L, double* rec, const int* __restrict__ idx, int n_envs, const float* act,
                                                       float* obs, unsigned char* done) {
  const int env = idx[blockIdx.x];
  if (env < 0 || env >= n_envs) return;
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  env_step_inner<T, RK ? 1 : 0>(M, K, L, rec + (size_t)env * L.stride, s, env, (const unsigned char*)nullptr, act, obs, done, (int)blockIdx.x);
}
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_physics(EnvRecordLayout L, double* rec,
                                                const double* ctrl, int nsub) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int env = blockIdx.x;
  env_physics<T, RK ? 1 : 0>(M, K, L, rec + (size_t)env * L.stride, s, env, ctrl, nsub);
}
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_dump(EnvRecordLayout L, double* rec, const double* ctrl,
                                             DumpLayout D, double* out) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int env = blockIdx.x;
  env_forward_dump<T>(M, K, L, rec + (size_t)env * L.stride, s, env, ctrl, D, out);
}
template <typename T, bool RK, int NC>
__global__ void __launch_bounds__(64, 2) k_wrap_census(EnvRecordLayout L, double* rec, int* cnt) {
  Scratch<T, NC>& s = *reinterpret_cast<Scratch<T, NC>*>(myo_lds);
  s.rk = rk_storage<T, RK, NC>();
  const DevModel<T>& M = myo_cmodel<T>();
  const TaskDev& K = c_task;
  const int env = blockIdx.x;
  env_wrap_census<T>(M, K, L, rec + (size_t)env * L.stride, s, env, cnt);
}
// ... and the order that follows from the census: positions sorted by count, most engaged first (ties keep their order), written back
// into the batch's own tables — gw_elem[k] = path element of position k, wr_i[8 w + 6] = position of element w — and the counts
// cleared.  One wave; ngw <= 128 (two positions per lane).  Each wrap's arithmetic is its own: the order changes no result bit.
__global__ void __launch_bounds__(64) k_wrap_reorder(int ngw, int* cnt, int* gw_elem, int* wr_i) {
  __shared__ int c[128], e[128];
  const int lane = threadIdx.x;
  for (int k = lane; k < 128; k += 64) { c[k] = k < ngw ? cnt[k] : -1; e[k] = k < ngw ? gw_elem[k] : 0; }
  __syncthreads();
  for (int k = lane; k < ngw; k += 64) {
    int rank = 0;
    for (int j = 0; j < ngw; ++j) rank += (c[j] > c[k]) || (c[j] == c[k] && j < k);
    gw_elem[rank] = e[k];
    wr_i[8 * e[k] + 6] = rank;
    cnt[k] = 0;
  }
}
// state gather/scatter: one thread per scalar
__global__ void k_state(double* rec, int stride, int off, int cnt, int n, double* ext, int to_ext) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * cnt) return;
  const int e = (int)(i / cnt), k = (int)(i % cnt);
  if (to_ext) ext[i] = rec[(size_t)e * stride + off + k]; else rec[(size_t)e * stride + off + k] = ext[i];
}
__global__ void k_state_i(double* rec, int stride, int off, int cnt, int n, int* ext, int to_ext) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * cnt) return;
  const int e = (int)(i / cnt), k = (int)(i % cnt);
  if (to_ext) ext[i] = (int)rec[(size_t)e * stride + off + k]; else rec[(size_t)e * stride + off + k] = (double)ext[i];
}
#define LAUNCH_CHECK(b)                                                                          \
  {                                                                                              \
    hipError_t _e = hipGetLastError();                                                           \
    if (_e != hipSuccess) return fail(MYO_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(_e)); \
  }
static void timing_begin(myo_batch* b, hipStream_t st) {
  if (!b->timing) return;
  hipEvent_t a, c;
  (void)hipEventCreate(&a); (void)hipEventCreate(&c);
  (void)hipEventRecord(a, st);
  b->pending.push_back({a, c});
}
static void timing_end(myo_batch* b, hipStream_t st) {
  if (!b->timing) return;
  (void)hipEventRecord(b->pending.back().second, st);
}

static unsigned lds_dyn(const myo_batch* b) {
#ifdef MYO_LDS_PAD_EXPERIMENT
  static int pad = -1;
  if (pad < 0) { const char* e = getenv("MYO_LDS_PAD"); pad = e ? atoi(e) : 0; }
  if (b->dtype != MYO_F64) return (unsigned)(MYO_LDS_ALIGN(sizeof(Scratch<float>)) + pad);
#endif
  return (unsigned)MYO_LDS_ALIGN(myo_batch_lds_bytes(b));
}
#define BIND_OR_RETURN(b, st) DeviceGuard _guard((b)->device); { int _rc = bind_constants(b, st); if (_rc) return _rc; }


// ---- the backend functions myo_host.h declares (myo_batch_create / _destroy call them)
static int be_batch_workspaces(myo_batch* b, int n_envs, int device) {
  (void)n_envs;
  int rc = 0;
    if (!slot_workspace_acquire(device, &b->K.ctrl_ws, &b->K.slot_map)) rc |= (int)hipErrorOutOfMemory; else b->has_slot_ws = true;
    if (!rc && b->ncap > MYO_NCON_MAX) {           // the 48-slot scratch keeps its contact records and wrap results in global memory (Scratch::SPILL)
      b->K.big_ws = big_workspace_acquire(device);
      if (!b->K.big_ws) rc |= (int)hipErrorOutOfMemory; else b->has_big_ws = true;
    }
  return rc;
}
static int be_batch_launch_state(myo_batch* b, const myo_model* m, int n_envs, int rc) {
  if (!rc) { rc |= (int)hipEventCreate(&b->ev0); rc |= (int)hipEventCreate(&b->ev1); }
  {
    // launch order of the env steps (k_step_order): opt-in, MYO_STEP_ORDER=1 — worth 3 % on whole-step launches, nothing once the
    // steps run in parts (2.239 vs 2.234 ms), where its 1024-thread sort per step costs what it gains
    const char* e = getenv("MYO_STEP_ORDER");
    if (!rc && b->K.kind && e && e[0] == '1') {
      std::vector<int> ident(n_envs);
      for (int i = 0; i < n_envs; ++i) ident[i] = i;
      void *po = nullptr, *pc = nullptr, *pt = nullptr;
      rc |= be_malloc(&po, sizeof(int) * (size_t)n_envs); rc |= be_malloc(&pc, sizeof(float) * (size_t)n_envs); rc |= be_malloc(&pt, sizeof(unsigned) * (size_t)n_envs);
      if (po) b->allocs.push_back(po);
      if (pc) b->allocs.push_back(pc);
      if (pt) b->allocs.push_back(pt);
      if (!rc) { rc |= be_h2d(po, ident.data(), sizeof(int) * (size_t)n_envs); rc |= (int)hipMemset(pc, 0, sizeof(float) * (size_t)n_envs); rc |= (int)hipMemset(pt, 0, sizeof(unsigned) * (size_t)n_envs); }
      if (!rc) { b->order = (int*)po; b->cost = (float*)pc; b->ticks = (unsigned*)pt; }
    }
    if (!rc && b->plan.nparts >= 2) {
      void *ps = nullptr, *pg = nullptr;
      rc |= be_malloc(&ps, sizeof(int) * (size_t)n_envs); rc |= be_malloc(&pg, sizeof(int) * 4);
      if (ps) b->allocs.push_back(ps);
      if (pg) b->allocs.push_back(pg);
      const int zero[4] = {0, 0, 0, 0};
      if (!rc) { rc |= (int)hipMemset(ps, 0, sizeof(int) * (size_t)n_envs); rc |= be_h2d(pg, zero, sizeof zero); }
      if (!rc) { b->part_state = (int*)ps; b->step_gen = (int*)pg; } else b->plan.nparts = 1;
    }
    if (!rc && m->ngw > 64 && !getenv("MYO_NO_WRAP_ORDER")) {        // (more than one pass of the wrap solver: worth ordering, see env_wrap_census)
      void* pw = nullptr;
      if (!be_malloc(&pw, sizeof(int) * (size_t)m->ngw)) {
        b->allocs.push_back(pw);
        if (hipMemset(pw, 0, sizeof(int) * (size_t)m->ngw) == hipSuccess) b->wrap_cnt = (int*)pw;
      }
    }
  }
  return rc;
}
static void be_batch_release(myo_batch* b, int device, int destroying) {
  if (!destroying) {      // a failed myo_batch_create: only the shared workspaces were taken
    if (b->has_slot_ws) slot_workspace_release(device);
    if (b->has_big_ws) big_workspace_release(device);
    return;
  }
  unbind(b, b->device);
  DeviceGuard guard(b->device);
  for (auto& pr : b->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1);
  if (b->has_slot_ws) { (void)hipDeviceSynchronize(); slot_workspace_release(b->device); }
  if (b->has_big_ws) big_workspace_release(b->device);
}

static void xfer(myo_batch* b, int off, int cnt, double* ext, int to_ext, be_stream st) {
  if (!ext) return;
  DeviceGuard guard(b->device);
  const long long tot = (long long)b->n * cnt;
  hipLaunchKernelGGL(k_state, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b->rec, b->L.stride, off, cnt, b->n, ext, to_ext);
}
static void xfer_i(myo_batch* b, int off, int cnt, int* ext, int to_ext, be_stream st) {
  if (!ext) return;
  DeviceGuard guard(b->device);
  const long long tot = (long long)b->n * cnt;
  hipLaunchKernelGGL(k_state_i, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b->rec, b->L.stride, off, cnt, b->n, ext, to_ext);
}

extern "C" int myo_batch_get_state(myo_batch* b, double* qpos, double* qvel, double* act, double* time, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer(b, b->L.off_qpos, b->nq, qpos, 1, st); xfer(b, b->L.off_qvel, b->nv, qvel, 1, st);
  xfer(b, b->L.off_act, b->na, act, 1, st); xfer(b, b->L.off_time, 1, time, 1, st);
  LAUNCH_CHECK(b)
  return MYO_OK;
}
extern "C" int myo_batch_set_state(myo_batch* b, const double* qpos, const double* qvel, const double* act,
                                   const double* time, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer(b, b->L.off_qpos, b->nq, (double*)qpos, 0, st); xfer(b, b->L.off_qvel, b->nv, (double*)qvel, 0, st);
  xfer(b, b->L.off_act, b->na, (double*)act, 0, st); xfer(b, b->L.off_time, 1, (double*)time, 0, st);
  LAUNCH_CHECK(b)
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
  LAUNCH_CHECK(b)
  return MYO_OK;
}
extern "C" int myo_batch_set_task(myo_batch* b, const int32_t* task_i, const double* task_d, const double* ball_d, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer_i(b, b->L.off_misc, 2, (int*)task_i, 0, st);
  xfer(b, b->L.off_taskd, MYO_TASKD_N, (double*)task_d, 0, st); xfer(b, b->L.off_balld, MYO_BALLD_N, (double*)ball_d, 0, st);
  LAUNCH_CHECK(b)
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
    (void)hipStreamSynchronize((hipStream_t)0);
    be_free(tmp);
  }
  unbind(b, b->device);              // the task block in __constant__ memory is re-uploaded at the next launch
  return MYO_OK;
}
extern "C" int myo_batch_object_friction(myo_batch* b, const double* set_fric, double* get_fric, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (b->K.objg_gidn <= 0) return fail(MYO_E_STATE, "batch has no object group");
  be_stream st = (be_stream)stream;
  const int cnt = 3 * (b->K.objg_gidn - b->K.objg_gid0);
  xfer(b, b->L.off_objfric, cnt, (double*)set_fric, 0, st); xfer(b, b->L.off_objfric, cnt, get_fric, 1, st);
  LAUNCH_CHECK(b)
  return MYO_OK;
}
extern "C" int myo_batch_bind_constants(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  DeviceGuard guard(b->device);
  return bind_constants(b, (hipStream_t)stream);
}
extern "C" int myo_batch_get_task(myo_batch* b, int32_t* task_i, double* task_d, double* ball_d, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  be_stream st = (be_stream)stream;
  xfer_i(b, b->L.off_misc, 2, (int*)task_i, 1, st);
  xfer(b, b->L.off_taskd, MYO_TASKD_N, task_d, 1, st); xfer(b, b->L.off_balld, MYO_BALLD_N, ball_d, 1, st);
  LAUNCH_CHECK(b)
  return MYO_OK;
}

// the RK4 stage storage is only allocated (LDS) by the kernel variants of RK4 models
// ... and the scratch's contact capacity (NCV) is the batch's: MYO_NCON_BIG for models with extended collision pairs or a die
#define MYO_NC_D(n) ((n) == MYO_NCON_MAX ? MYO_NCON_F64 : (n))     /* the fp64 stepper's scratch has its own base capacity */
#define LAUNCH_RK1(b, ...) if ((b)->integrator == 1) { constexpr bool RKV = true; __VA_ARGS__; } else { constexpr bool RKV = false; __VA_ARGS__; }
#define LAUNCH_RK(b, ...) if ((b)->ncap > MYO_NCON_MAX) { constexpr int NCV = MYO_NCON_BIG; LAUNCH_RK1(b, __VA_ARGS__) } else { constexpr int NCV = MYO_NCON_MAX; LAUNCH_RK1(b, __VA_ARGS__) }

// census of the wraps over the envs' present states, then the new order (both on the stream: usable between any two steps, also while
// a graph of steps exists — the tables are rewritten in place)
static void wrap_order_launch(myo_batch* b, hipStream_t st) {
  if (!b->wrap_cnt) return;
  int* gw = const_cast<int*>(b->dtype == MYO_F64 ? b->Md.gw_elem.p : b->Mf.gw_elem.p);
  int* wr = const_cast<int*>(b->dtype == MYO_F64 ? b->Md.wr_i.p : b->Mf.wr_i.p);
  const int ngw = b->dtype == MYO_F64 ? b->Md.ngw : b->Mf.ngw;
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wrap_census<double, RKV, MYO_NC_D(NCV)>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, b->wrap_cnt);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wrap_census<float, RKV, NCV>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, b->wrap_cnt))
  hipLaunchKernelGGL(k_wrap_reorder, dim3(1), dim3(64), 0, st, ngw, b->wrap_cnt, gw, wr);
}
extern "C" int myo_batch_tune_wrap_order(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  wrap_order_launch(b, st);
  LAUNCH_CHECK(b)
  return MYO_OK;
}

extern "C" int myo_batch_reset(myo_batch* b, const uint8_t* mask, float* obs, void* stream) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reset<double, RKV, MYO_NC_D(NCV)>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, mask, obs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reset<float, RKV, NCV>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, mask, obs))
  LAUNCH_CHECK(b)
  if (!mask) { wrap_order_launch(b, st); b->wrap_tune_in = 16; LAUNCH_CHECK(b) }       // a reset of every env: the wrap order from the reset states, again 16 steps on
  return MYO_OK;
}

extern "C" int myo_batch_step(myo_batch* b, const float* act, float* obs, float* rew, uint8_t* done, uint8_t* trunc,
                              float* term_obs, float* comps, float* ep_info, void* stream) {
  if (!b || !act || !obs || !rew || !done) return fail(MYO_E_ARG, "myo_batch_step: act/obs/rew/done are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  timing_begin(b, st);
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step<double, RKV, MYO_NC_D(NCV)>), dim3(b->plan.nparts * b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, act, obs, rew, done, trunc, term_obs, comps, ep_info, b->bad_state, b->order, b->ticks, b->part_state, b->step_gen, b->plan);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step<float, RKV, NCV>), dim3(b->plan.nparts * b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, act, obs, rew, done, trunc, term_obs, comps, ep_info, b->bad_state, b->order, b->ticks, b->part_state, b->step_gen, b->plan))
  timing_end(b, st);
  LAUNCH_CHECK(b)                 // the generation below only advances behind a k_step that was launched
  if (b->wrap_cnt && --b->wrap_tune_in <= 0) {       // the wrap order follows the states the envs are in (env_wrap_census)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap == hipStreamCaptureStatusNone) { wrap_order_launch(b, st); b->wrap_tune_in = 256; }      // (never baked into a captured graph of steps)
    else b->wrap_tune_in = 1;
  }
  if (b->order || b->step_gen) hipLaunchKernelGGL(k_step_order, dim3(1), dim3(b->order ? 1024 : 64), 0, st, (const unsigned int*)b->ticks, b->cost, b->n, b->order, b->step_gen);
  LAUNCH_CHECK(b)
  return MYO_OK;
}

// test hook: put the step plan's generation counter (and every env's state, consistently) at `gen` — the counter wraps after
// 2^28 steps and the protocol's arithmetic is modulo 2^32 (tests/test_step_parts.py steps across the wrap)
extern "C" int myo_batch_set_step_generation(myo_batch* b, unsigned int gen) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  if (!b->part_state) return MYO_OK;
  DeviceGuard guard(b->device);
  std::vector<int> st((size_t)b->n, (int)(16u * gen));
  const int g4[4] = {(int)gen, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess || be_h2d(b->part_state, st.data(), sizeof(int) * st.size()) || be_h2d(b->step_gen, g4, sizeof g4))
    return fail(MYO_E_DEVICE, "myo_batch_set_step_generation: copy failed");
  return MYO_OK;
}

// Health counters of a batch (synchronises the device).  out[0]: k_step workgroups that found their env's hand-off state in another
// generation than the launch's (see the protocol comment at k_step); out[1]: substeps in which an env had more contacts than its
// scratch holds (the surplus was dropped); out[2]: the same for limit rows; out[3]: the most contact slots such a substep asked for.  All 0 in a healthy batch.
extern "C" int myo_batch_health(myo_batch* b, int out[4]) {
  if (!b || !out) return fail(MYO_E_ARG, "null argument");
  for (int k = 0; k < 4; ++k) out[k] = 0;
  if (!b->K.health) return MYO_OK;
  DeviceGuard guard(b->device);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, b->K.health, 4 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(MYO_E_DEVICE, "myo_batch_health: copy failed");
  return MYO_OK;
}

// Device check of what the per-slot workspace rests on (myo_wave_slot, wave.h): `n_workgroups` one-wave workgroups with `lds_bytes` of
// dynamic LDS (k_step's footprint: the same residency) each take their slot's occupancy counter, stay for ~20 us, and leave.
// out[0]: workgroups that found their slot occupied (must be 0); out[1]: distinct slots seen; out[2]: largest slot index; out[3]: bit mask of XCC ids.
__global__ void __launch_bounds__(64, 2) k_wave_slot_probe(int* occ, int* seen, int* out) {
  if (threadIdx.x != 0) return;
  const unsigned slot = myo_wave_slot(0);
  if (atomicAdd(occ + slot, 1) != 0) atomicAdd(out, 1);
  if (atomicExch(seen + slot, 1) == 0) atomicAdd(out + 1, 1);
  atomicMax(out + 2, (int)slot);
  atomicOr(out + 3, 1 << (slot >> 14));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000ull) __builtin_amdgcn_s_sleep(8);
  atomicSub(occ + slot, 1);
}
extern "C" int myo_debug_wave_slots(int device, int n_workgroups, int lds_bytes, int32_t out[4]) {
  if (!out || n_workgroups <= 0 || lds_bytes < 0 || lds_bytes > 65536) return fail(MYO_E_ARG, "bad argument");
  DeviceGuard guard(device);
  int *occ = nullptr, *seen = nullptr, *o = nullptr;
  const size_t nb = sizeof(int) * (size_t)MYO_WAVE_SLOTS;
  int rc = (int)hipMalloc((void**)&occ, nb) | (int)hipMalloc((void**)&seen, nb) | (int)hipMalloc((void**)&o, 4 * sizeof(int));
  if (!rc) rc = (int)hipMemset(occ, 0, nb) | (int)hipMemset(seen, 0, nb) | (int)hipMemset(o, 0, 4 * sizeof(int));
  if (!rc) {
    if (lds_bytes > 0) (void)hipFuncSetAttribute((const void*)k_wave_slot_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k_wave_slot_probe, dim3(n_workgroups), dim3(64), (unsigned)lds_bytes, 0, occ, seen, o);
    rc = (int)hipGetLastError() | (int)hipDeviceSynchronize();
  }
  if (!rc) rc = (int)hipMemcpy(out, o, 4 * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(occ); (void)hipFree(seen); (void)hipFree(o);
  if (rc) return fail(MYO_E_DEVICE, "myo_debug_wave_slots: %s", hipGetErrorString((hipError_t)rc));
  return MYO_OK;
}

extern "C" int myo_batch_step_inner(myo_batch* b, const uint8_t* mask, const float* act, float* obs, uint8_t* done, void* stream) {
  if (!b || !act || !obs) return fail(MYO_E_ARG, "myo_batch_step_inner: act/obs are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_inner<double, RKV, MYO_NC_D(NCV)>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, mask, act, obs, done);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_inner<float, RKV, NCV>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, mask, act, obs, done))
  LAUNCH_CHECK(b)
  return MYO_OK;
}

// Compact form of myo_batch_step_inner: the envs idx[0 .. n_idx) (dev int32; -1 = empty slot) take one unwrapped env step with
// row r of act [n_idx, nu]; row r of obs [n_idx, obs_dim] and done [n_idx] (may be NULL) receive env idx[r]'s results.  An env
// must not be listed twice.  MixtureModelBaodingEnv's base phase runs the few envs that were just reset this way.
extern "C" int myo_batch_step_inner_idx(myo_batch* b, const int* idx, int n_idx, const float* act, float* obs, uint8_t* done, void* stream) {
  if (!b || !idx || !act || !obs || n_idx <= 0) return fail(MYO_E_ARG, "myo_batch_step_inner_idx: idx/act/obs are required");
  if (!b->K.kind) return fail(MYO_E_STATE, "batch has no task layer");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_inner_idx<double, RKV, MYO_NC_D(NCV)>), dim3(n_idx), dim3(64), lds_dyn(b), st, b->L, b->rec, idx, b->n, act, obs, done);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_inner_idx<float, RKV, NCV>), dim3(n_idx), dim3(64), lds_dyn(b), st, b->L, b->rec, idx, b->n, act, obs, done))
  LAUNCH_CHECK(b)
  return MYO_OK;
}

// Whole env records from one batch into another (same model, same task kind): dst env dst_idx[r] <- src env src_idx[r], r < k.
// The record is everything an env is between two steps (state, warm start, task scalars, per-episode draws, counters), so the
// destination env continues exactly where the source env stood.  MixtureModelBaodingEnv hands pre-played episodes (reset + base
// phase, done in bulk on a pool batch) to the envs that have just finished.
__global__ void k_copy_envs(double* dst, const int* __restrict__ dst_idx, int n_dst, const double* src, const int* __restrict__ src_idx, int n_src, int stride) {
  const int d = dst_idx[blockIdx.x], s = src_idx[blockIdx.x];
  if (d < 0 || d >= n_dst || s < 0 || s >= n_src) return;
  for (int i = threadIdx.x; i < stride; i += blockDim.x) dst[(size_t)d * stride + i] = src[(size_t)s * stride + i];
}
extern "C" int myo_batch_copy_envs(myo_batch* dst, const int* dst_idx, const myo_batch* src, const int* src_idx, int k, void* stream) {
  if (!dst || !src || !dst_idx || !src_idx || k < 0) return fail(MYO_E_ARG, "myo_batch_copy_envs: null argument");
  if (dst->L.stride != src->L.stride || dst->nq != src->nq || dst->nv != src->nv || dst->na != src->na || dst->K.kind != src->K.kind || dst->device != src->device)
    return fail(MYO_E_ARG, "myo_batch_copy_envs: the two batches differ in model, task kind or device");
  if (k == 0) return MYO_OK;
  DeviceGuard guard(dst->device);
  hipLaunchKernelGGL(k_copy_envs, dim3(k), dim3(64), 0, (hipStream_t)stream, dst->rec, dst_idx, dst->n, src->rec, src_idx, src->n, dst->L.stride);
  LAUNCH_CHECK(dst)
  return MYO_OK;
}

extern "C" int myo_batch_physics_step(myo_batch* b, const double* ctrl, int nsub, void* stream) {
  if (!b || nsub < 0) return fail(MYO_E_ARG, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  timing_begin(b, st);
  LAUNCH_RK(b,
    if (b->dtype == MYO_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_physics<double, RKV, MYO_NC_D(NCV)>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, ctrl, nsub);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_physics<float, RKV, NCV>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, ctrl, nsub))
  timing_end(b, st);
  LAUNCH_CHECK(b)
  return MYO_OK;
}

extern "C" int myo_batch_forward_dump(myo_batch* b, const double* ctrl, double* out, void* stream) {
  if (!b || !out) return fail(MYO_E_ARG, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  BIND_OR_RETURN(b, st)
  LAUNCH_RK(b, (void)RKV;
    if (b->dtype == MYO_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dump<double, false, MYO_NC_D(NCV)>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, ctrl, b->D, out);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dump<float, false, NCV>), dim3(b->n), dim3(64), lds_dyn(b), st, b->L, b->rec, ctrl, b->D, out))
  LAUNCH_CHECK(b)
  return MYO_OK;
}

extern "C" int myo_batch_enable_timing(myo_batch* b, int on) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  b->timing = on;
  return MYO_OK;
}
extern "C" double myo_batch_kernel_ms(myo_batch* b) {
  if (!b) return -1.0;
  double sum = 0;
  int cnt = 0;
  for (auto& pr : b->pending) {
    (void)hipEventSynchronize(pr.second);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { sum += ms; cnt++; }
    (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
  }
  b->pending.clear();
  return cnt ? sum / cnt : -1.0;
}


// ------------------------------------------------------------------------------------------ PPO loss
// Elementwise part of one PPO minibatch step (SB3 PPO.train semantics) in one launch:
// Gaussian log-prob, ratio, clipped surrogate, value loss and their gradients.  A block owns 64
// consecutive rows: the [64,A] tiles of mean / actions are staged through LDS with coalesced loads
// (row stride A is odd for the hand, A = 39, so the per-row reads are conflict-free), dmean leaves
// the same way.  acc[2A+3] = {d loss/d log_std[A] (without the entropy term), pl, vl,
// sum_i dmean[i,:] (bias grad of the action head), sum_i dvalue[i] (bias grad of the value head)}.
__device__ __forceinline__ unsigned short myo_f2bf(float x) {   // round-to-nearest-even, as torch's .to(bfloat16)
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <bool IN_BF16>
__global__ void __launch_bounds__(64) k_ppo_loss(const float* __restrict__ mean, const float* __restrict__ values,
                                                 const float* __restrict__ actions, const float* __restrict__ old_logp,
                                                 const float* __restrict__ adv, const float* __restrict__ returns,
                                                 const float* __restrict__ log_std, const float* __restrict__ adv_stats,
                                                 int B, int A, float clip, float vf_coef, float* __restrict__ dmean,
                                                 float* __restrict__ dvalue, float* __restrict__ acc,
                                                 unsigned short* __restrict__ dmean_h, unsigned short* __restrict__ dvalue_h,
                                                 float* __restrict__ part) {
  __shared__ float t_mean[64 * 64], t_act[64 * 64], s_ls[64], s_inv[64];
  const int lane = threadIdx.x, r0 = blockIdx.x * 64;
  const int rows = (B - r0) < 64 ? (B - r0) : 64;
  const int n = rows * A;
  const size_t base = (size_t)r0 * A;
  const unsigned short* mean_h = reinterpret_cast<const unsigned short*>(mean);
  for (int k = lane; k < n; k += 64) {
    t_mean[k] = IN_BF16 ? __uint_as_float(((unsigned)mean_h[base + k]) << 16) : mean[base + k];
    t_act[k] = actions[base + k];
  }
  if (lane < A) { const float ls = log_std[lane]; s_ls[lane] = ls; s_inv[lane] = __expf(-ls); }
  __syncthreads();
  const int i = r0 + lane;
  const bool on = lane < rows;
  float pl_i = 0.f, vl_i = 0.f, dlogp = 0.f, dv_i = 0.f;
  if (on) {
    float logp = 0.f;
    for (int a = 0; a < A; ++a) {
      const float z = (t_act[lane * A + a] - t_mean[lane * A + a]) * s_inv[a];
      logp += -0.5f * z * z - s_ls[a] - 0.9189385332046727f;
    }
    const float an = (adv[i] - adv_stats[0]) / (adv_stats[1] + 1e-8f);
    const float ratio = __expf(logp - old_logp[i]);
    const float s1 = an * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float s2 = an * rc;
    pl_i = -fminf(s1, s2) / B;
    const bool inside = (ratio > 1.f - clip) && (ratio < 1.f + clip);
    dlogp = -(an * ratio) * ((s1 <= s2) ? 1.f : (inside ? 1.f : 0.f)) / B;
    const float val = IN_BF16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(values)[i]) << 16) : values[i];
    const float dv = val - returns[i];
    vl_i = dv * dv / B;
    dv_i = vf_coef * 2.f / B * dv;
    dvalue[i] = dv_i;
    if (dvalue_h) dvalue_h[i] = myo_f2bf(dv_i);
  }
  // per-row gradient tiles: dmean into t_mean, dlogp (z^2-1) into t_act (own row only: no hazard)
  if (on) {
    for (int a = 0; a < A; ++a) {
      const float inv = s_inv[a];
      const float z = (t_act[lane * A + a] - t_mean[lane * A + a]) * inv;
      t_mean[lane * A + a] = dlogp * z * inv;
      t_act[lane * A + a] = dlogp * (z * z - 1.f);
    }
  }
  __syncthreads();
  float my_ls = 0.f, my_db = 0.f;     // lane a: column sums of action dim a over the block's rows
  if (lane < A)
    for (int r = 0; r < rows; ++r) { my_ls += t_act[r * A + lane]; my_db += t_mean[r * A + lane]; }
  for (int k = lane; k < n; k += 64) {
    const float v = t_mean[k];
    dmean[base + k] = v;
    if (dmean_h) dmean_h[base + k] = myo_f2bf(v);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    pl_i += __shfl_xor(pl_i, off, 64); vl_i += __shfl_xor(vl_i, off, 64); dv_i += __shfl_xor(dv_i, off, 64);
  }
  // per-block partial sums (column-major: part[c, block]); k_colmajor_finish adds them up
  // in a fixed order (deterministic, no float atomics)
  const int W = 2 * A + 3, NB = gridDim.x;
  if (lane < A) { part[(size_t)lane * NB + blockIdx.x] = my_ls; part[(size_t)(A + 2 + lane) * NB + blockIdx.x] = my_db; }
  if (lane == 0) {
    part[(size_t)A * NB + blockIdx.x] = pl_i; part[(size_t)(A + 1) * NB + blockIdx.x] = vl_i;
    part[(size_t)(2 * A + 2) * NB + blockIdx.x] = dv_i;
  }
}
// acc[c] = sum over blocks of part[c, block] (fixed order: deterministic); one wave per column.
// A separate launch instead of a last-block epilogue: an in-kernel release fence writes back the
// whole L2 of the XCD on this chip (measured: +90 us), a kernel boundary is cheaper.
__global__ void __launch_bounds__(64) k_colmajor_finish(const float* __restrict__ part, float* __restrict__ acc, int NB, int A,
                                                        float ent_coef, float* __restrict__ g_log_std,
                                                        float* __restrict__ g_bias_pi, float* __restrict__ g_bias_vf) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float a = 0.f;
  for (int b = lane; b < NB; b += 64) a += part[(size_t)c * NB + b];
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if (lane == 0) {
    acc[c] = a;
    if (g_log_std && c < A) g_log_std[c] = a - ent_coef;          // d(-ent_coef * entropy)/d log_std = -ent_coef
    if (g_bias_pi && c >= A + 2 && c < 2 * A + 2) g_bias_pi[c - A - 2] = a;
    if (g_bias_vf && c == 2 * A + 2) g_bias_vf[0] = a;
  }
}
extern "C" int myo_ppo_loss_grad(const float* mean, const float* values, const float* actions, const float* old_logp,
                                 const float* adv, const float* returns, const float* log_std, const float* adv_stats,
                                 int B, int A, float clip, float vf_coef, float* dmean, float* dvalue, float* acc,
                                 uint16_t* dmean_bf16, uint16_t* dvalue_bf16, float* work, int in_bf16, float ent_coef,
                                 float* g_log_std, float* g_bias_pi, float* g_bias_vf, void* stream) {
  if (!mean || !values || !actions || !old_logp || !adv || !returns || !log_std || !adv_stats || !dmean || !dvalue || !acc ||
      !work || B <= 0 || A <= 0 || A > 64)
    return fail(MYO_E_ARG, "myo_ppo_loss_grad: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (B + 63) / 64;
  if (in_bf16)
    hipLaunchKernelGGL(k_ppo_loss<true>, dim3(nblk), dim3(64), 0, st, mean, values, actions, old_logp, adv, returns,
                       log_std, adv_stats, B, A, clip, vf_coef, dmean, dvalue, acc, dmean_bf16, dvalue_bf16, work);
  else
    hipLaunchKernelGGL(k_ppo_loss<false>, dim3(nblk), dim3(64), 0, st, mean, values, actions, old_logp, adv, returns,
                       log_std, adv_stats, B, A, clip, vf_coef, dmean, dvalue, acc, dmean_bf16, dvalue_bf16, work);
  hipLaunchKernelGGL(k_colmajor_finish, dim3(2 * A + 3), dim3(64), 0, st, work, acc, nblk, A, ent_coef, g_log_std, g_bias_pi,
                     g_bias_vf);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ split-K
// out[g, j] = sum_k part[g, k, j]: the reduction that finishes a split-K weight-gradient bmm
// (bf16 partial products, fp32 sum).  torch's generic reduce over a middle dimension takes 37 us on
// a [2,32,256,256] tensor; this is one coalesced pass.
template <bool BF16>
__global__ void __launch_bounds__(256) k_splitk_reduce(const unsigned* __restrict__ part, float2* __restrict__ out, int splits,
                                                       int n2, int total2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;   // index of an element pair in the output
  if (j >= total2) return;
  const int g = j / n2, c = j - g * n2;
  float a0 = 0.f, a1 = 0.f;
  if (BF16) {
    const unsigned* src = part + (size_t)g * splits * n2 + c;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) {
      const unsigned u = src[(size_t)k * n2];
      a0 += __uint_as_float(u << 16);
      a1 += __uint_as_float(u & 0xffff0000u);
    }
  } else {
    const float2* src = reinterpret_cast<const float2*>(part) + (size_t)g * splits * n2 + c;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) { const float2 v = src[(size_t)k * n2]; a0 += v.x; a1 += v.y; }
  }
  out[j] = make_float2(a0, a1);
}
// dy <- dy * (act > 0) (ReLU backward, bf16 in place) and per-block column sums of the result:
// partial[blk, c] over the block's 32 rows.  cols/2 must divide 256 (cols = 256: one column pair
// per thread, two row phases).  Followed by k_splitk_reduce<false> for the bias gradient.
__global__ void __launch_bounds__(256) k_relu_bwd_colsum(unsigned* __restrict__ dy, const unsigned* __restrict__ act, int P,
                                                         float2* __restrict__ partial) {
  __shared__ float2 red[256];
  const int t = threadIdx.x, cp = t % P, rph = t / P, nph = 256 / P;
  const size_t row0 = (size_t)blockIdx.x * 32;
  float a0 = 0.f, a1 = 0.f;
  for (int r = rph; r < 32; r += nph) {
    const size_t i = (row0 + r) * P + cp;
    const unsigned g = dy[i], a = act[i];
    // bf16 > 0  <=>  sign bit clear and magnitude non-zero
    const unsigned lo = ((a & 0x8000u) == 0 && (a & 0x7fffu) != 0) ? (g & 0xffffu) : 0u;
    const unsigned hi = ((a & 0x80000000u) == 0 && (a & 0x7fff0000u) != 0) ? (g & 0xffff0000u) : 0u;
    dy[i] = lo | hi;
    a0 += __uint_as_float(lo << 16);
    a1 += __uint_as_float(hi);
  }
  red[t] = make_float2(a0, a1);
  __syncthreads();
  if (t < P) {
    float2 v = red[t];
    for (int k = 1; k < nph; ++k) { v.x += red[t + k * P].x; v.y += red[t + k * P].y; }
    partial[(size_t)blockIdx.x * P + t] = v;
  }
}
// h <- max(h + bias, 0) in place, bf16 [G, R, C] with bias [G, C] (the epilogue of a batched GEMM).
__global__ void __launch_bounds__(256) k_bias_relu(unsigned* __restrict__ h, const unsigned* __restrict__ bias, int P, size_t RP,
                                                   size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t g = i / RP;
  const int cp = (int)(i % P);
  const unsigned x = h[i], b = bias[g * P + cp];
  const float lo = fmaxf(__uint_as_float(x << 16) + __uint_as_float(b << 16), 0.f);
  const float hi = fmaxf(__uint_as_float(x & 0xffff0000u) + __uint_as_float(b & 0xffff0000u), 0.f);
  h[i] = (unsigned)myo_f2bf(lo) | ((unsigned)myo_f2bf(hi) << 16);
}
// Minibatch gather of one PPO optimiser step: rows idx[0..bs) of the rollout arrays -> contiguous
// minibatch arrays (obs as `copies` stacked bf16 copies: the GEMM operand of the stacked actor /
// critic trunks), plus mean and unbiased std of the gathered advantages (block-wise mean / M2, merged
// by k_moments_finish in block order: deterministic).  Replaces 5 index kernels, 3 casts/copies and
// the var_mean reduction.
#define MYO_GATHER_ROWS 16
__global__ void __launch_bounds__(256) k_ppo_gather(const float* __restrict__ obs, const float* __restrict__ act,
                                                    const float* __restrict__ oldlp, const float* __restrict__ adv,
                                                    const float* __restrict__ ret, const long long* __restrict__ idx, int bs,
                                                    int O, int A, unsigned short* __restrict__ obs_h, int copies,
                                                    float* __restrict__ act_mb, float* __restrict__ oldlp_mb,
                                                    float* __restrict__ adv_mb, float* __restrict__ ret_mb,
                                                    float* __restrict__ part) {
  constexpr int R = MYO_GATHER_ROWS;
  __shared__ long long s_idx[R];
  const int t = threadIdx.x, r0 = blockIdx.x * R;
  const int rows = (bs - r0) < R ? (bs - r0) : R;
  if (t < rows) s_idx[t] = idx[r0 + t];
  __syncthreads();
#pragma unroll 2
  for (int e = t; e < rows * O; e += 256) {
    const int r = e / O, c = e - r * O;
    const unsigned short v = myo_f2bf(obs[(size_t)s_idx[r] * O + c]);
    for (int k = 0; k < copies; ++k) obs_h[((size_t)k * bs + r0 + r) * O + c] = v;
  }
#pragma unroll 2
  for (int e = t; e < rows * A; e += 256) {
    const int r = e / A, c = e - r * A;
    act_mb[(size_t)(r0 + r) * A + c] = act[(size_t)s_idx[r] * A + c];
  }
  if (t < 64) {
    float a = 0.f;
    if (t < rows) {
      const size_t src = (size_t)s_idx[t];
      a = adv[src];
      oldlp_mb[r0 + t] = oldlp[src]; adv_mb[r0 + t] = a; ret_mb[r0 + t] = ret[src];
    }
    float sum = a;
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / rows;
    const float d = (t < rows) ? (a - mean) : 0.f;
    float m2 = d * d;
    for (int off = 32; off >= 1; off >>= 1) m2 += __shfl_xor(m2, off, 64);
    if (t == 0) { part[2 * blockIdx.x] = mean; part[2 * blockIdx.x + 1] = m2; }
  }
}
// merges the block moments of k_ppo_gather in block order (Chan): mean and unbiased std
__global__ void __launch_bounds__(64) k_moments_finish(const float* __restrict__ part, int nb, int bs, float* __restrict__ adv_stats) {
  constexpr int R = MYO_GATHER_ROWS;
  const int t = threadIdx.x;
  float wsum = 0.f;
  for (int b = t; b < nb; b += 64) {
    const int n_b = (bs - b * R) < R ? (bs - b * R) : R;
    wsum += n_b * part[2 * b];
  }
  for (int off = 32; off >= 1; off >>= 1) wsum += __shfl_xor(wsum, off, 64);
  const float mean = wsum / bs;
  float m2 = 0.f;
  for (int b = t; b < nb; b += 64) {
    const int n_b = (bs - b * R) < R ? (bs - b * R) : R;
    const float dm = part[2 * b] - mean;
    m2 += part[2 * b + 1] + n_b * dm * dm;
  }
  for (int off = 32; off >= 1; off >>= 1) m2 += __shfl_xor(m2, off, 64);
  if (t == 0) { adv_stats[0] = mean; adv_stats[1] = sqrtf(m2 / (bs > 1 ? bs - 1 : 1)); }
}
// Tall fp32 reduction out[g, c] = sum_k part[g, k, c] for few columns and many splits (bias gradients):
// a block owns 32 column pairs, 8 split phases per pair, combined through LDS in phase order.
__global__ void __launch_bounds__(256) k_colsum_finish(const float2* __restrict__ part, float2* __restrict__ out, int splits, int n2) {
  __shared__ float2 red[256];
  const int t = threadIdx.x, cl = t & 31, ph = t >> 5;
  const int j = blockIdx.x * 32 + cl;                 // output pair index (over groups * n2)
  const int g = j / n2, c = j - g * n2;
  const float2* src = part + (size_t)g * splits * n2 + c;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
  for (int k = ph; k < splits; k += 8) { const float2 v = src[(size_t)k * n2]; a0 += v.x; a1 += v.y; }
  red[t] = make_float2(a0, a1);
  __syncthreads();
  if (ph == 0) {
    float2 v = red[cl];
    for (int k = 1; k < 8; ++k) { v.x += red[cl + 32 * k].x; v.y += red[cl + 32 * k].y; }
    out[j] = v;
  }
}
// Two independent split reductions in ONE launch (the weight-gradient partials of a layer and its bias
// partials; the two head gradients): blocks [0, nbA) run job A, the rest job B, each with the code of the
// stand-alone kernels above (mode 0: bf16 partials, 1: tall fp32 through LDS, 2: fp32 partials).
struct ReduceJob { const void* part; float2* out; int splits, n2, total2, mode, nblocks; };
__device__ __forceinline__ void myo_reduce_job(const ReduceJob& J, int vb, float2* red) {
  const int t = threadIdx.x;
  if (J.mode == 1) {
    const int cl = t & 31, ph = t >> 5;
    const int j = vb * 32 + cl;
    const int g = j / J.n2, c = j - g * J.n2;
    const float2* src = reinterpret_cast<const float2*>(J.part) + (size_t)g * J.splits * J.n2 + c;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
    for (int k = ph; k < J.splits; k += 8) { const float2 v = src[(size_t)k * J.n2]; a0 += v.x; a1 += v.y; }
    red[t] = make_float2(a0, a1);
    __syncthreads();
    if (ph == 0) {
      float2 v = red[cl];
      for (int k = 1; k < 8; ++k) { v.x += red[cl + 32 * k].x; v.y += red[cl + 32 * k].y; }
      J.out[j] = v;
    }
    return;
  }
  const int j = vb * 256 + t;
  if (j >= J.total2) return;
  const int g = j / J.n2, c = j - g * J.n2;
  float a0 = 0.f, a1 = 0.f;
  if (J.mode == 0) {
    const unsigned* src = reinterpret_cast<const unsigned*>(J.part) + (size_t)g * J.splits * J.n2 + c;
#pragma unroll 8
    for (int k = 0; k < J.splits; ++k) {
      const unsigned u = src[(size_t)k * J.n2];
      a0 += __uint_as_float(u << 16);
      a1 += __uint_as_float(u & 0xffff0000u);
    }
  } else {
    const float2* src = reinterpret_cast<const float2*>(J.part) + (size_t)g * J.splits * J.n2 + c;
#pragma unroll 8
    for (int k = 0; k < J.splits; ++k) { const float2 v = src[(size_t)k * J.n2]; a0 += v.x; a1 += v.y; }
  }
  J.out[j] = make_float2(a0, a1);
}
__global__ void __launch_bounds__(256) k_reduce_pair(ReduceJob A, ReduceJob B) {
  __shared__ float2 red[256];
  if ((int)blockIdx.x < A.nblocks) myo_reduce_job(A, blockIdx.x, red);      // block-uniform branch
  else myo_reduce_job(B, blockIdx.x - A.nblocks, red);
}
extern "C" int myo_bias_relu_bf16(uint16_t* h, const uint16_t* bias, int groups, int rows, int cols, void* stream) {
  if (!h || !bias || groups <= 0 || rows <= 0 || cols <= 0 || (cols & 1)) return fail(MYO_E_ARG, "myo_bias_relu_bf16: bad arguments");
  const size_t P = cols / 2, RP = (size_t)rows * P, total = RP * groups;
  hipLaunchKernelGGL(k_bias_relu, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned*)h,
                     (const unsigned*)bias, (int)P, RP, total);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_ppo_gather(const float* obs, const float* act, const float* oldlp, const float* adv, const float* ret,
                              const int64_t* idx, int bs, int obs_dim, int act_dim, uint16_t* obs_bf16, int copies,
                              float* act_mb, float* oldlp_mb, float* adv_mb, float* ret_mb, float* adv_stats, float* work,
                              void* stream) {
  if (!obs || !act || !oldlp || !adv || !ret || !idx || !obs_bf16 || !act_mb || !oldlp_mb || !adv_mb || !ret_mb ||
      !work || bs <= 0 || obs_dim <= 0 || act_dim <= 0 || copies <= 0)
    return fail(MYO_E_ARG, "myo_ppo_gather: bad arguments");
  hipLaunchKernelGGL(k_ppo_gather, dim3((bs + MYO_GATHER_ROWS - 1) / MYO_GATHER_ROWS), dim3(256), 0, (hipStream_t)stream, obs, act, oldlp, adv, ret,
                     (const long long*)idx, bs, obs_dim, act_dim, obs_bf16, copies, act_mb, oldlp_mb, adv_mb, ret_mb, work);
  if (adv_stats)      // NULL: the caller supplies the advantage moments itself (cross-rank moments, or no normalisation)
    hipLaunchKernelGGL(k_moments_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, work,
                       (bs + MYO_GATHER_ROWS - 1) / MYO_GATHER_ROWS, bs, adv_stats);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

static ReduceJob make_reduce_job(const void* part, int part_is_bf16, float* out, int groups, int splits, int n) {
  ReduceJob J;
  J.part = part; J.out = (float2*)out; J.splits = splits; J.n2 = n / 2; J.total2 = groups * (n / 2);
  if (part_is_bf16) { J.mode = 0; J.nblocks = (J.total2 + 255) / 256; }
  else if (J.n2 % 32 == 0 && splits >= 64) { J.mode = 1; J.nblocks = J.total2 / 32; }
  else { J.mode = 2; J.nblocks = (J.total2 + 255) / 256; }
  return J;
}
extern "C" int myo_splitk_reduce2(const void* part_a, int a_is_bf16, float* out_a, int groups_a, int splits_a, int n_a,
                                  const void* part_b, int b_is_bf16, float* out_b, int groups_b, int splits_b, int n_b, void* stream) {
  if (!part_a || !out_a || groups_a <= 0 || splits_a <= 0 || n_a <= 0 || (n_a & 1) || !part_b || !out_b || groups_b <= 0 ||
      splits_b <= 0 || n_b <= 0 || (n_b & 1))
    return fail(MYO_E_ARG, "myo_splitk_reduce2: bad arguments");
  const ReduceJob A = make_reduce_job(part_a, a_is_bf16, out_a, groups_a, splits_a, n_a);
  const ReduceJob B = make_reduce_job(part_b, b_is_bf16, out_b, groups_b, splits_b, n_b);
  hipLaunchKernelGGL(k_reduce_pair, dim3(A.nblocks + B.nblocks), dim3(256), 0, (hipStream_t)stream, A, B);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_splitk_reduce(const void* part, int part_is_bf16, float* out, int groups, int splits, int n, void* stream) {
  if (!part || !out || groups <= 0 || splits <= 0 || n <= 0 || (n & 1)) return fail(MYO_E_ARG, "myo_splitk_reduce: bad arguments");
  const int total2 = groups * (n / 2);
  if (part_is_bf16)
    hipLaunchKernelGGL(k_splitk_reduce<true>, dim3((total2 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)part, (float2*)out, splits, n / 2, total2);
  else if ((n / 2) % 32 == 0 && splits >= 64)
    hipLaunchKernelGGL(k_colsum_finish, dim3(total2 / 32), dim3(256), 0, (hipStream_t)stream, (const float2*)part, (float2*)out,
                       splits, n / 2);
  else
    hipLaunchKernelGGL(k_splitk_reduce<false>, dim3((total2 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)part, (float2*)out, splits, n / 2, total2);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_relu_bwd_colsum_bf16(uint16_t* dy, const uint16_t* act, int rows, int cols, float* partial, void* stream) {
  if (!dy || !act || !partial || rows <= 0 || cols <= 0 || (rows % 32) || (cols & 1) || (256 % (cols / 2)) || cols / 2 < 8)
    return fail(MYO_E_ARG, "myo_relu_bwd_colsum_bf16: rows must be a multiple of 32 and cols/2 a divisor of 256 (>= 8)");
  hipLaunchKernelGGL(k_relu_bwd_colsum, dim3(rows / 32), dim3(256), 0, (hipStream_t)stream, (unsigned*)dy, (const unsigned*)act,
                     cols / 2, (float2*)partial);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ rollout
// The per-step plumbing between two env kernels (policy input, action sampling, VecNormalize, rollout
// buffer writes) as a handful of launches instead of ~80 small framework kernels.  `t_idx` is a
// DEVICE int32 holding the rollout-buffer row of the current step, so the same launches replay from a
// hipGraph for every step.
// obs (f32 [N,O]) -> obs_buf[t] (f32) and `copies` stacked bf16 copies (GEMM operand of the trunks)
__global__ void __launch_bounds__(256) k_obs_prepare(const float* __restrict__ obs, size_t n, float* __restrict__ obs_buf,
                                                     unsigned short* __restrict__ x_h, int copies, const int* __restrict__ t_idx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = obs[i];
  if (obs_buf) obs_buf[(size_t)(*t_idx) * n + i] = v;
  const unsigned short h = myo_f2bf(v);
  for (int k = 0; k < copies; ++k) x_h[(size_t)k * n + i] = h;
}
// DiagGaussian sampling (SB3 DiagGaussianDistribution: a = mu + exp(log_std) eps; log-prob summed over
// dims) for N rows, one wave per 64 rows x loop over action dims; Philox4x32-10 keyed by (seed, rollout
// step counter), Box-Muller.  Writes act_buf[t], val_buf[t], logp_buf[t] and the clipped actions.
// 16 lanes per env, lane g = the four actions 4g..4g+3 (one Philox block each, counter (env, g, draw)): the rows of act_buf /
// clipped are written 16 B per lane, contiguous across the 16 lanes, and log pi is a 16-lane shuffle sum.  (One lane per env
// walked the ten Philox blocks of a 39-action row serially and stored with a 156 B stride: 12 us for 4096 envs.)
#define MYO_SAMPLE_ROWS 16
__global__ void __launch_bounds__(16 * MYO_SAMPLE_ROWS) k_sample_actions(const unsigned short* __restrict__ mean_h, const unsigned short* __restrict__ value_h,
                                                       const float* __restrict__ log_std, int N, int A, unsigned long long seed,
                                                       unsigned long long* __restrict__ draw_counter, const int* __restrict__ t_idx,
                                                       float* __restrict__ act_buf, float* __restrict__ val_buf,
                                                       float* __restrict__ logp_buf, float* __restrict__ clipped, int deterministic) {
  const int i = blockIdx.x * MYO_SAMPLE_ROWS + (threadIdx.x >> 4), g0 = threadIdx.x & 15;
  const size_t t = (size_t)(*t_idx);
  const unsigned long long ctr = *draw_counter;
  float logp = 0.f;
  if (i < N) {
    for (int a0 = 4 * g0; a0 < A; a0 += 64) {
      unsigned int c[4] = {(unsigned int)i, (unsigned int)(a0 >> 2), (unsigned int)ctr, (unsigned int)(ctr >> 32)};
      unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
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
          const float ls = log_std[a];
          const float mu = __uint_as_float(((unsigned)mean_h[(size_t)i * A + a]) << 16);
          const float act = deterministic ? mu : mu + __expf(ls) * z[k];
          const float zz = (act - mu) * __expf(-ls);
          logp += -0.5f * zz * zz - ls - 0.9189385332046727f;
          act_buf[(t * N + i) * A + a] = act;
          clipped[(size_t)i * A + a] = fminf(fmaxf(act, -1.f), 1.f);
        }
      }
    }
  }
  for (int off = 8; off >= 1; off >>= 1) logp += __shfl_xor(logp, off, 16);
  if (i < N && g0 == 0) {
    logp_buf[t * N + i] = logp;
    val_buf[t * N + i] = __uint_as_float(((unsigned)value_h[i]) << 16);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) draw_counter[1] = ctr + 1;   // committed by k_rollout_advance
}
// VecNormalize.step_wait (SB3 1.6.2), kernel 1 of 3: per-block fp64 moments of the raw
// observation columns over 128-row chunks, discounted-return update ret = ret*gamma + r and its moments.
#define MYO_VN_ROWS 128
__global__ void __launch_bounds__(128) k_vecnorm_moments(const float* __restrict__ obs, const float* __restrict__ rew, int N, int O,
                                                         double* __restrict__ returns, double gamma, int training,
                                                         double* __restrict__ part) {
  __shared__ double red[128];
  const int t = threadIdx.x, r0 = blockIdx.x * MYO_VN_ROWS;
  const int rows = (N - r0) < MYO_VN_ROWS ? (N - r0) : MYO_VN_ROWS;
  double* mine = part + (size_t)blockIdx.x * 2 * (O + 1);
  // column c of the chunk: thread pair (2c', 2c'+1) -> each half walks alternate rows with 4 independent
  // accumulators (the loads of a pass are in flight together), halves meet through a lane shuffle
  const int half = t & 1;
  for (int c = t >> 1; c < O; c += 64) {
    const float* col = obs + (size_t)r0 * O + c;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int r = half;
#pragma unroll 4
    for (; r + 6 < rows; r += 8) {
      a0 += (double)col[(size_t)r * O]; a1 += (double)col[(size_t)(r + 2) * O];
      a2 += (double)col[(size_t)(r + 4) * O]; a3 += (double)col[(size_t)(r + 6) * O];
    }
    for (; r < rows; r += 2) a0 += (double)col[(size_t)r * O];
    double sum = (a0 + a1) + (a2 + a3);
    sum += __shfl_xor(sum, 1);
    const double mean = sum / rows;
    a0 = a1 = a2 = a3 = 0;
    r = half;
#pragma unroll 4
    for (; r + 6 < rows; r += 8) {
      const double d0 = (double)col[(size_t)r * O] - mean, d1 = (double)col[(size_t)(r + 2) * O] - mean;
      const double d2 = (double)col[(size_t)(r + 4) * O] - mean, d3 = (double)col[(size_t)(r + 6) * O] - mean;
      a0 += d0 * d0; a1 += d1 * d1; a2 += d2 * d2; a3 += d3 * d3;
    }
    for (; r < rows; r += 2) { const double d = (double)col[(size_t)r * O] - mean; a0 += d * d; }
    double m2 = (a0 + a1) + (a2 + a3);
    m2 += __shfl_xor(m2, 1);
    if (!half) { mine[2 * c] = mean; mine[2 * c + 1] = m2; }
  }
  double v = 0;
  if (t < rows) {
    v = returns[r0 + t];
    if (training) { v = v * gamma + (double)rew[r0 + t]; returns[r0 + t] = v; }
  }
  red[t] = (t < rows) ? v : 0.0;
  __syncthreads();
  for (int sft = 64; sft >= 1; sft >>= 1) { if (t < sft) red[t] += red[t + sft]; __syncthreads(); }
  const double mean = red[0] / rows;
  __syncthreads();
  const double d = (t < rows) ? (v - mean) : 0.0;
  red[t] = d * d;
  __syncthreads();
  for (int sft = 64; sft >= 1; sft >>= 1) { if (t < sft) red[t] += red[t + sft]; __syncthreads(); }
  if (t == 0) { mine[2 * O] = mean; mine[2 * O + 1] = red[0]; }
}
// kernel 2 of 3 (one block): merge the block moments in block order -> batch mean / variance, then the
// running statistics by Chan's parallel update (RunningMeanStd.update_from_moments), in place.
__global__ void __launch_bounds__(128) k_vecnorm_merge(const double* __restrict__ part, int nb, int N, int O, double* __restrict__ obs_mean,
                                                       double* __restrict__ obs_var, double* __restrict__ obs_count,
                                                       double* __restrict__ ret_stats, int upd_obs, int upd_ret) {
  const int t = threadIdx.x;
  const double oc = *obs_count, rc = ret_stats[2];
  __syncthreads();
  for (int c = t; c <= O; c += 128) {
    if ((c < O && !upd_obs) || (c == O && !upd_ret)) continue;
    // (the block moments are fetched eight blocks at a time, all loads of a group in flight before the first add: one
    // dependent L2 round trip per block was 13 us for the 32 blocks of 4096 envs; the additions keep their order)
    double wsum = 0;
    for (int b0 = 0; b0 < nb; b0 += 8) {
      double pm[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pm[k] = (b0 + k < nb) ? part[((size_t)(b0 + k) * (O + 1) + c) * 2] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + k;
        const int n_b = (N - b * MYO_VN_ROWS) < MYO_VN_ROWS ? (N - b * MYO_VN_ROWS) : MYO_VN_ROWS;
        if (b < nb) wsum += n_b * pm[k];
      }
    }
    const double bmean = wsum / N;
    double m2 = 0;
    for (int b0 = 0; b0 < nb; b0 += 8) {
      double pm[8], pv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        pm[k] = (b0 + k < nb) ? part[((size_t)(b0 + k) * (O + 1) + c) * 2] : 0.0;
        pv[k] = (b0 + k < nb) ? part[((size_t)(b0 + k) * (O + 1) + c) * 2 + 1] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + k;
        const int n_b = (N - b * MYO_VN_ROWS) < MYO_VN_ROWS ? (N - b * MYO_VN_ROWS) : MYO_VN_ROWS;
        const double dm = pm[k] - bmean;
        if (b < nb) m2 += pv[k] + n_b * dm * dm;
      }
    }
    const double bvar = m2 / N;
    double* mean = c < O ? obs_mean + c : ret_stats;
    double* var = c < O ? obs_var + c : ret_stats + 1;
    const double cnt = c < O ? oc : rc;
    const double delta = bmean - *mean, tot = cnt + N;
    const double new_mean = *mean + delta * N / tot;
    const double M2 = *var * cnt + bvar * N + delta * delta * cnt * N / tot;
    *mean = new_mean; *var = M2 / tot;
  }
  if (t == 0) { if (upd_obs) *obs_count = oc + N; if (upd_ret) ret_stats[2] = rc + N; }
}
// ---- the same update split at the point where N ranks exchange their batch moments (all-reduce
// (n, sum x, sum x^2) so that N ranks reproduce ONE VecNormalize over all envs).  k_vecnorm_batch turns the block
// moments of this rank into additive sums batch[0] = n, batch[1 + c] = sum x_c, batch[2 + O + c] = sum x_c^2 (column
// O = the discounted returns); after the caller's all-reduce(SUM) k_vecnorm_merge_batch does Chan's update from them.
__global__ void __launch_bounds__(128) k_vecnorm_batch(const double* __restrict__ part, int nb, int N, int O, double* __restrict__ batch) {
  const int t = threadIdx.x;
  for (int c = t; c <= O; c += 128) {
    double s1 = 0, s2 = 0;
    for (int b = 0; b < nb; ++b) {
      const int n_b = (N - b * MYO_VN_ROWS) < MYO_VN_ROWS ? (N - b * MYO_VN_ROWS) : MYO_VN_ROWS;
      const double m = part[((size_t)b * (O + 1) + c) * 2];
      s1 += n_b * m;
      s2 += part[((size_t)b * (O + 1) + c) * 2 + 1] + n_b * m * m;
    }
    batch[1 + c] = s1; batch[2 + O + c] = s2;
  }
  if (t == 0) batch[0] = (double)N;
}
__global__ void __launch_bounds__(128) k_vecnorm_merge_batch(const double* __restrict__ batch, int O, double* __restrict__ obs_mean,
                                                             double* __restrict__ obs_var, double* __restrict__ obs_count,
                                                             double* __restrict__ ret_stats, int upd_obs, int upd_ret) {
  const int t = threadIdx.x;
  const double oc = *obs_count, rc = ret_stats[2], n = batch[0];
  __syncthreads();
  for (int c = t; c <= O; c += 128) {
    if ((c < O && !upd_obs) || (c == O && !upd_ret)) continue;
    const double bmean = batch[1 + c] / n;
    double bvar = batch[2 + O + c] / n - bmean * bmean;
    bvar = bvar < 0 ? 0 : bvar;
    double* mean = c < O ? obs_mean + c : ret_stats;
    double* var = c < O ? obs_var + c : ret_stats + 1;
    const double cnt = c < O ? oc : rc;
    const double delta = bmean - *mean, tot = cnt + n;
    const double new_mean = *mean + delta * n / tot;
    const double M2 = *var * cnt + bvar * n + delta * delta * cnt * n / tot;
    *mean = new_mean; *var = M2 / tot;
  }
  if (t == 0) { if (upd_obs) *obs_count = oc + n; if (upd_ret) ret_stats[2] = rc + n; }
}
// kernel 3 of 3: normalise observation, terminal observation and reward with the UPDATED statistics,
// zero the returns of finished episodes, and write the rollout buffer rows of step t.
__global__ void __launch_bounds__(256) k_vecnorm_apply(const float* __restrict__ obs, const float* __restrict__ rew,
                                                       const unsigned char* __restrict__ done, const unsigned char* __restrict__ trunc,
                                                       const float* __restrict__ term, int N, int O, const double* __restrict__ obs_mean,
                                                       const double* __restrict__ obs_var, const double* __restrict__ ret_stats,
                                                       double* __restrict__ returns, double eps, double clip_obs, double clip_rew,
                                                       int norm_obs, int norm_rew, float* __restrict__ nobs, float* __restrict__ starts,
                                                       const int* __restrict__ t_idx, float* __restrict__ rew_buf,
                                                       float* __restrict__ start_buf, float* __restrict__ term_buf,
                                                       float* __restrict__ trunc_buf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)N * O, tt = (size_t)(*t_idx);
  if (i < n) {
    const int c = (int)(i % O);
    double o = obs[i], q = term[i];
    if (norm_obs) {
      const double sd = sqrt(obs_var[c] + eps), mu = obs_mean[c];
      o = fmin(fmax((o - mu) / sd, -clip_obs), clip_obs);
      q = fmin(fmax((q - mu) / sd, -clip_obs), clip_obs);
    }
    nobs[i] = (float)o;
    if (term_buf) term_buf[tt * n + i] = (float)q;
  }
  if (i < (size_t)N) {
    double r = rew[i];
    if (norm_rew) r = fmin(fmax(r / sqrt(ret_stats[1] + eps), -clip_rew), clip_rew);
    rew_buf[tt * N + i] = (float)r;
    start_buf[tt * N + i] = starts[i];          // episode_starts of the step that produced this transition
    const unsigned char dn = done[i];
    starts[i] = dn ? 1.f : 0.f;
    if (trunc_buf) trunc_buf[tt * N + i] = trunc[i] ? 1.f : 0.f;
    if (dn) returns[i] = 0.0;
  }
}
// gSDE action sampling (SB3 StateDependentNoiseDistribution): one 64-lane block per env, lane j =
// action j.  noise_j = sum_l latent_l W[n][l][j] with the env's own exploration matrix W (drawn once per rollout),
// sigma_j = sqrt(sum_l latent_l^2 exp(log_std[l][j])^2 + 1e-6); action = mean + noise (unclipped, as stored by SB3),
// clipped copy for the env, log pi = sum_j log N(action_j; mean_j, sigma_j).  W rows are read coalesced (A contiguous).
__global__ void __launch_bounds__(64) k_sample_sde(const float* __restrict__ mean, const float* __restrict__ latent,
                                                   const float* __restrict__ W, const float* __restrict__ log_std, int N, int L, int A,
                                                   float* __restrict__ actions, float* __restrict__ clipped, float* __restrict__ logp,
                                                   int deterministic) {
  const int n = blockIdx.x, t = threadIdx.x;
  float lp_part = 0.f;
  for (int j = t; j < A; j += 64) {
    const float* w = W + (size_t)n * L * A + j;
    const float* lat = latent + (size_t)n * L;
    float noise = 0.f, var = 0.f;
    for (int l = 0; l < L; ++l) {
      const float x = lat[l], sd = __expf(log_std[(size_t)l * A + j]);
      noise = fmaf(x, w[(size_t)l * A], noise);
      var = fmaf(x * x, sd * sd, var);
    }
    const float mu = mean[(size_t)n * A + j], sg = sqrtf(var + 1e-6f);
    const float a = deterministic ? mu : mu + noise;
    actions[(size_t)n * A + j] = a;
    clipped[(size_t)n * A + j] = fminf(fmaxf(a, -1.f), 1.f);
    const float z = (a - mu) / sg;
    lp_part += -0.5f * z * z - logf(sg) - 0.9189385332046727f;
  }
  for (int off = 32; off >= 1; off >>= 1) lp_part += __shfl_xor(lp_part, off, 64);
  if (t == 0) logp[n] = lp_part;
}
__global__ void k_rollout_advance(int* t_idx, int T, unsigned long long* draw_counter) {
  *t_idx = (*t_idx + 1) % T;
  draw_counter[0] = draw_counter[1];
}
extern "C" int myo_rollout_policy_input(const float* obs, int N, int O, float* obs_buf, uint16_t* x_bf16, int copies,
                                        const int32_t* t_idx, void* stream) {
  if (!obs || !x_bf16 || !t_idx || N <= 0 || O <= 0 || copies <= 0) return fail(MYO_E_ARG, "myo_rollout_policy_input: bad arguments");
  const size_t n = (size_t)N * O;
  hipLaunchKernelGGL(k_obs_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, n, obs_buf, x_bf16, copies, t_idx);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_rollout_sample(const uint16_t* mean_bf16, const uint16_t* value_bf16, const float* log_std, int N, int A,
                                  uint64_t seed, uint64_t* draw_counter, const int32_t* t_idx, float* act_buf, float* val_buf,
                                  float* logp_buf, float* clipped, int deterministic, void* stream) {
  if (!mean_bf16 || !value_bf16 || !log_std || !draw_counter || !t_idx || !act_buf || !val_buf || !logp_buf || !clipped || N <= 0 || A <= 0)
    return fail(MYO_E_ARG, "myo_rollout_sample: bad arguments");
  hipLaunchKernelGGL(k_sample_actions, dim3((N + MYO_SAMPLE_ROWS - 1) / MYO_SAMPLE_ROWS), dim3(16 * MYO_SAMPLE_ROWS), 0, (hipStream_t)stream, mean_bf16, value_bf16, log_std, N, A,
                     (unsigned long long)seed, (unsigned long long*)draw_counter, t_idx, act_buf, val_buf, logp_buf, clipped, deterministic);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_vecnorm_step(const float* obs, const float* rew, const uint8_t* done, const uint8_t* trunc, const float* term_obs,
                                int N, int O, double* obs_mean, double* obs_var, double* obs_count, double* ret_stats,
                                double* returns, double gamma, double eps, double clip_obs, double clip_rew, int training,
                                int norm_obs, int norm_reward, float* nobs, float* starts, const int32_t* t_idx, float* rew_buf,
                                float* start_buf, float* term_buf, float* trunc_buf, double* work, void* stream) {
  if (!obs || !rew || !done || !trunc || !term_obs || !obs_mean || !obs_var || !obs_count || !ret_stats || !returns || !nobs ||
      !starts || !t_idx || !rew_buf || !start_buf || !work || N <= 0 || O <= 0)
    return fail(MYO_E_ARG, "myo_vecnorm_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (N + MYO_VN_ROWS - 1) / MYO_VN_ROWS;
  hipLaunchKernelGGL(k_vecnorm_moments, dim3(nb), dim3(128), 0, st, obs, rew, N, O, returns, gamma, training, work);
  if (training)
    hipLaunchKernelGGL(k_vecnorm_merge, dim3(1), dim3(128), 0, st, work, nb, N, O, obs_mean, obs_var, obs_count, ret_stats,
                       (int)(training && norm_obs), (int)(training != 0));
  const size_t n = (size_t)N * O;
  hipLaunchKernelGGL(k_vecnorm_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, obs, rew, done, trunc, term_obs, N, O,
                     obs_mean, obs_var, ret_stats, returns, eps, clip_obs, clip_rew, norm_obs, norm_reward, nobs, starts, t_idx,
                     rew_buf, start_buf, term_buf, trunc_buf);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_rollout_sample_sde(const float* mean, const float* latent, const float* exploration_mat, const float* log_std,
                                      int N, int L, int A, float* actions, float* clipped, float* logp, int deterministic, void* stream) {
  if (!mean || !latent || !exploration_mat || !log_std || !actions || !clipped || !logp || N <= 0 || L <= 0 || A <= 0)
    return fail(MYO_E_ARG, "myo_rollout_sample_sde: bad arguments");
  hipLaunchKernelGGL(k_sample_sde, dim3(N), dim3(64), 0, (hipStream_t)stream, mean, latent, exploration_mat, log_std, N, L, A, actions,
                     clipped, logp, deterministic);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_vecnorm_batch_moments(const float* obs, const float* rew, int N, int O, double* returns, double gamma, int training,
                                         double* work, double* batch, void* stream) {
  if (!obs || !rew || !returns || !work || !batch || N <= 0 || O <= 0) return fail(MYO_E_ARG, "myo_vecnorm_batch_moments: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (N + MYO_VN_ROWS - 1) / MYO_VN_ROWS;
  hipLaunchKernelGGL(k_vecnorm_moments, dim3(nb), dim3(128), 0, st, obs, rew, N, O, returns, gamma, training, work);
  hipLaunchKernelGGL(k_vecnorm_batch, dim3(1), dim3(128), 0, st, work, nb, N, O, batch);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_vecnorm_finish(const float* obs, const float* rew, const uint8_t* done, const uint8_t* trunc, const float* term_obs,
                                  int N, int O, double* obs_mean, double* obs_var, double* obs_count, double* ret_stats,
                                  double* returns, double eps, double clip_obs, double clip_rew, int training, int norm_obs,
                                  int norm_reward, float* nobs, float* starts, const int32_t* t_idx, float* rew_buf, float* start_buf,
                                  float* term_buf, float* trunc_buf, const double* batch, void* stream) {
  if (!obs || !rew || !done || !trunc || !term_obs || !obs_mean || !obs_var || !obs_count || !ret_stats || !returns || !nobs ||
      !starts || !t_idx || !rew_buf || !start_buf || !batch || N <= 0 || O <= 0)
    return fail(MYO_E_ARG, "myo_vecnorm_finish: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (training)
    hipLaunchKernelGGL(k_vecnorm_merge_batch, dim3(1), dim3(128), 0, st, batch, O, obs_mean, obs_var, obs_count, ret_stats,
                       (int)(training && norm_obs), (int)(training != 0));
  const size_t n = (size_t)N * O;
  hipLaunchKernelGGL(k_vecnorm_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, obs, rew, done, trunc, term_obs, N, O,
                     obs_mean, obs_var, ret_stats, returns, eps, clip_obs, clip_rew, norm_obs, norm_reward, nobs, starts, t_idx,
                     rew_buf, start_buf, term_buf, trunc_buf);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_rollout_advance(int32_t* t_idx, int T, uint64_t* draw_counter, void* stream) {
  if (!t_idx || !draw_counter || T <= 0) return fail(MYO_E_ARG, "myo_rollout_advance: bad arguments");
  hipLaunchKernelGGL(k_rollout_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, t_idx, T, (unsigned long long*)draw_counter);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ LSTM cell
// One time step of G stacked one-layer LSTMs (PyTorch gate order i, f, g, o) around the recurrent GEMM,
// forward and backward (RecurrentActorCriticPolicy's lstm_actor / lstm_critic).  Rows
// r = g * N + n.  The episode-start mask of the NEXT step is folded in: the forward kernel emits the unmasked
// h (the sequence output) and the masked h, c that feed the next step's GEMM / cell; the backward kernel
// applies the same mask to the incoming state gradients.  E = float or bf16 storage, fp32 arithmetic.
template <typename E> __device__ __forceinline__ float myo_ld(const E* p, size_t i);
template <> __device__ __forceinline__ float myo_ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float myo_ld<unsigned short>(const unsigned short* p, size_t i) { return __uint_as_float((unsigned)p[i] << 16); }
template <typename E> __device__ __forceinline__ void myo_st(E* p, size_t i, float v);
template <> __device__ __forceinline__ void myo_st<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void myo_st<unsigned short>(unsigned short* p, size_t i, float v) { p[i] = myo_f2bf(v); }
__device__ __forceinline__ float myo_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
template <typename E>
__global__ void __launch_bounds__(256) k_lstm_cell_fwd(const E* __restrict__ gx, const E* __restrict__ gh, const E* __restrict__ c_prev,
                                                       const float* __restrict__ keep_next, int R, int N, int H, E* __restrict__ out_h,
                                                       E* __restrict__ hm_next, E* __restrict__ cm_next, E* __restrict__ c_new,
                                                       E* __restrict__ ws) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)R * H) return;
  const int r = (int)(idx / H), j = (int)(idx % H);
  const size_t g0 = (size_t)r * 4 * H + j;
  const float i = myo_sigmoid(myo_ld(gx, g0) + myo_ld(gh, g0));
  const float f = myo_sigmoid(myo_ld(gx, g0 + H) + myo_ld(gh, g0 + H));
  const float g = tanhf(myo_ld(gx, g0 + 2 * H) + myo_ld(gh, g0 + 2 * H));
  const float o = myo_sigmoid(myo_ld(gx, g0 + 3 * H) + myo_ld(gh, g0 + 3 * H));
  const float c = f * myo_ld(c_prev, idx) + i * g;
  const float h = o * tanhf(c);
  const float k = keep_next ? keep_next[r % N] : 1.f;
  myo_st(out_h, idx, h);
  myo_st(c_new, idx, c);
  myo_st(hm_next, idx, h * k);
  myo_st(cm_next, idx, c * k);
  myo_st(ws, g0, i); myo_st(ws, g0 + H, f); myo_st(ws, g0 + 2 * H, g); myo_st(ws, g0 + 3 * H, o);
}
template <typename E>
__global__ void __launch_bounds__(256) k_lstm_cell_bwd(const E* __restrict__ dout, const E* __restrict__ dhm_next, const E* __restrict__ dcm_next,
                                                       const float* __restrict__ keep_next, const E* __restrict__ c_prev,
                                                       const E* __restrict__ c_new, const E* __restrict__ ws, int R, int N, int H,
                                                       E* __restrict__ dgates, E* __restrict__ dc_prev) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)R * H) return;
  const int r = (int)(idx / H), j = (int)(idx % H);
  const size_t g0 = (size_t)r * 4 * H + j;
  const float k = keep_next ? keep_next[r % N] : 1.f;
  const float dh = (dout ? myo_ld(dout, idx) : 0.f) + (dhm_next ? k * myo_ld(dhm_next, idx) : 0.f);
  const float dc_in = dcm_next ? k * myo_ld(dcm_next, idx) : 0.f;
  const float i = myo_ld(ws, g0), f = myo_ld(ws, g0 + H), g = myo_ld(ws, g0 + 2 * H), o = myo_ld(ws, g0 + 3 * H);
  const float tc = tanhf(myo_ld(c_new, idx));
  const float dct = dc_in + dh * o * (1.f - tc * tc);
  myo_st(dgates, g0, dct * g * i * (1.f - i));
  myo_st(dgates, g0 + H, dct * myo_ld(c_prev, idx) * f * (1.f - f));
  myo_st(dgates, g0 + 2 * H, dct * i * (1.f - g * g));
  myo_st(dgates, g0 + 3 * H, dh * tc * o * (1.f - o));
  myo_st(dc_prev, idx, dct * f);
}
extern "C" int myo_lstm_cell_fwd(const void* gx, const void* gh, const void* c_prev, const float* keep_next, int R, int N, int H,
                                 int is_bf16, void* out_h, void* hm_next, void* cm_next, void* c_new, void* ws, void* stream) {
  if (!gx || !gh || !c_prev || !out_h || !hm_next || !cm_next || !c_new || !ws || R <= 0 || N <= 0 || H <= 0 || R % N)
    return fail(MYO_E_ARG, "myo_lstm_cell_fwd: bad arguments");
  const unsigned nb = (unsigned)(((size_t)R * H + 255) / 256);
  if (is_bf16)
    hipLaunchKernelGGL(k_lstm_cell_fwd<unsigned short>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gx,
                       (const unsigned short*)gh, (const unsigned short*)c_prev, keep_next, R, N, H, (unsigned short*)out_h,
                       (unsigned short*)hm_next, (unsigned short*)cm_next, (unsigned short*)c_new, (unsigned short*)ws);
  else
    hipLaunchKernelGGL(k_lstm_cell_fwd<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)gx, (const float*)gh,
                       (const float*)c_prev, keep_next, R, N, H, (float*)out_h, (float*)hm_next, (float*)cm_next, (float*)c_new,
                       (float*)ws);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_lstm_cell_bwd(const void* dout, const void* dhm_next, const void* dcm_next, const float* keep_next,
                                 const void* c_prev, const void* c_new, const void* ws, int R, int N, int H, int is_bf16,
                                 void* dgates, void* dc_prev, void* stream) {
  if (!c_prev || !c_new || !ws || !dgates || !dc_prev || R <= 0 || N <= 0 || H <= 0 || R % N)
    return fail(MYO_E_ARG, "myo_lstm_cell_bwd: bad arguments");
  const unsigned nb = (unsigned)(((size_t)R * H + 255) / 256);
  if (is_bf16)
    hipLaunchKernelGGL(k_lstm_cell_bwd<unsigned short>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dout,
                       (const unsigned short*)dhm_next, (const unsigned short*)dcm_next, keep_next, (const unsigned short*)c_prev,
                       (const unsigned short*)c_new, (const unsigned short*)ws, R, N, H, (unsigned short*)dgates,
                       (unsigned short*)dc_prev);
  else
    hipLaunchKernelGGL(k_lstm_cell_bwd<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)dout, (const float*)dhm_next,
                       (const float*)dcm_next, keep_next, (const float*)c_prev, (const float*)c_new, (const float*)ws, R, N, H,
                       (float*)dgates, (float*)dc_prev);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ GAE
// compute_returns_and_advantage of SB3's RolloutBuffer: backward scan over T with
// the episode_starts[t+1] mask; one thread per env, coalesced over envs.  [T,N] row-major.
__global__ void k_gae(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ starts,
                      const float* __restrict__ last_val, const float* __restrict__ last_done, int T, int N, float gamma,
                      float lam, float* __restrict__ adv, float* __restrict__ ret) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  float last = 0.f, nextv = last_val[e], nonterm = 1.f - last_done[e];
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * N + e;
    const float v = val[i];
    const float delta = rew[i] + gamma * nextv * nonterm - v;
    last = delta + gamma * lam * nonterm * last;
    adv[i] = last;
    ret[i] = last + v;
    nextv = v;
    nonterm = 1.f - starts[i];
  }
}
extern "C" int myo_gae(const float* rew, const float* val, const float* starts, const float* last_val, const float* last_done,
                       int T, int N, float gamma, float lam, float* adv, float* ret, void* stream) {
  if (!rew || !val || !starts || !last_val || !last_done || !adv || !ret || T <= 0 || N <= 0) return fail(MYO_E_ARG, "myo_gae: bad arguments");
  hipLaunchKernelGGL(k_gae, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, val, starts, last_val, last_done, T, N,
                     gamma, lam, adv, ret);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ Adam
// clip_grad_norm_(max_norm) + torch.optim.Adam step (no weight decay / amsgrad) over ONE flat fp32
// parameter vector: 2 launches instead of ~12 multi-tensor ones.  scratch[0] = sum g^2 (after
// grad_scale), step = device-side step counter (incremented here).  SB3 semantics: SURVEY.md C.5.
#define MYO_SQN_BLOCKS 64
// scratch[0..63] <- per-block sums of (g*gs)^2 (fixed order inside a block: no float atomics);
// step[0] <- committed step count, step[1] <- pending (= this step's ordinal)
__global__ void __launch_bounds__(256) k_grad_sqnorm(const float* __restrict__ g, int n, float gs, float* __restrict__ part,
                                                     int* __restrict__ step) {
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += MYO_SQN_BLOCKS * 256) { const float x = g[i] * gs; acc += x * x; }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0) { const int done = step[1]; step[0] = done; step[1] = done + 1; }
  }
}
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int n, float lr, float b1, float b2, float eps,
                                              float max_norm, float gs, const int* __restrict__ step,
                                              const float* __restrict__ part, unsigned short* __restrict__ p_bf16, int nparts) {
  // |g|^2 from the partial sums: every block adds them in the same tree order (256 lanes, strided, then halving) — a serial
  // loop over them was one dependent scalar load per partial, 4 us for 64 and 21 us for 337
  __shared__ float sq_red[256];
  {
    float a = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 256) a += part[k];
    sq_red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) { if ((int)threadIdx.x < w) sq_red[threadIdx.x] += sq_red[threadIdx.x + w]; __syncthreads(); }
  }
  const float sq = sq_red[0];
  const int t = step[1];
  const float norm = sqrtf(sq);
  const float clip = (max_norm > 0.f) ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  const float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float gi = g[i] * gs * clip;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pn = p[i] - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pn;
    if (p_bf16) p_bf16[i] = myo_f2bf(pn);          // the bf16 shadow the GEMMs read: no separate cast pass
  }
}
extern "C" int myo_adam_clip_step(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2,
                                  float eps, float max_norm, float grad_scale, int* step, float* scratch, uint16_t* p_bf16,
                                  void* stream) {
  if (!p || !g || !m || !v || !step || !scratch || n <= 0) return fail(MYO_E_ARG, "myo_adam_clip_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(k_grad_sqnorm, dim3(MYO_SQN_BLOCKS), dim3(256), 0, st, g, n, grad_scale, scratch, step);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, max_norm, grad_scale, step, scratch,
                     (unsigned short*)p_bf16, MYO_SQN_BLOCKS);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_adam_apply(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2,
                              float eps, float max_norm, float grad_scale, const int* step, const float* scratch, int nparts,
                              uint16_t* p_bf16, void* stream) {
  if (!p || !g || !m || !v || !step || !scratch || n <= 0 || nparts <= 0) return fail(MYO_E_ARG, "myo_adam_apply: bad arguments");
  const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, max_norm, grad_scale, step, scratch,
                     (unsigned short*)p_bf16, nparts);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ fused MLP PPO step
#include "myo_ppo_mlp.h"
struct MlpWs { unsigned short *W1p, *W2, *W2T, *Whp, *WhT; float* bias; unsigned short *XT, *H1T, *H2T, *dH1T, *dH2T, *dOT; float *part, *slab, *advpart; size_t bytes; };
static MlpWs mlp_carve(unsigned char* base, int B, int OP, int A, long long G) {
  MlpWs w;
  size_t o = 0;
  auto take = [&](size_t n) { unsigned char* p = base ? base + o : nullptr; o += (n + 255) / 256 * 256; return p; };
  const size_t H = MLP_H;
  w.W1p = (unsigned short*)take(2 * H * OP * 2); w.W2 = (unsigned short*)take(2 * H * H * 2); w.W2T = (unsigned short*)take(2 * H * H * 2);
  w.Whp = (unsigned short*)take(2 * MLP_APM * H * 2); w.WhT = (unsigned short*)take(2 * H * MLP_AKP * 2);
  w.bias = (float*)take((4 * H + 2 * MLP_APM) * 4);
  w.XT = (unsigned short*)take((size_t)128 * B * 2);
  w.H1T = (unsigned short*)take(2 * H * (size_t)B * 2); w.H2T = (unsigned short*)take(2 * H * (size_t)B * 2);
  w.dH1T = (unsigned short*)take(2 * H * (size_t)B * 2); w.dH2T = (unsigned short*)take(2 * H * (size_t)B * 2);
  w.dOT = (unsigned short*)take((size_t)256 * B * 2);
  w.part = (float*)take((size_t)(2 * A + 3) * (B / MLP_BM) * 4);
  w.slab = (float*)take((size_t)MLP_SPLITK * G * 4);
  w.advpart = (float*)take(3 * MLP_ADV_BLOCKS * 4);            // slices of the advantage moments
  w.bytes = o;
  return w;
}
static int mlp_shape_ok(int B, int O, int A, int hidden) {
  return hidden == MLP_H && O >= 1 && O <= MLP_OPMAX && A >= 1 && A <= MLP_APM && B >= 64 * MLP_SPLITK && B % (64 * MLP_SPLITK) == 0;
}
extern "C" long long myo_ppo_mlp_workspace_bytes(int B, int obs_dim, int act_dim, int hidden, long long G) {
  if (!mlp_shape_ok(B, obs_dim, act_dim, hidden) || G <= 0) return -1;
  return (long long)mlp_carve(nullptr, B, (obs_dim + 31) / 32 * 32, act_dim, G).bytes;
}
extern "C" int myo_ppo_mlp_sqnorm_parts(int act_dim) {
  return MLP_RF_BLOCKS + 2 * act_dim + 3;
}
extern "C" int myo_ppo_mlp_step(const myo_ppo_mlp_desc* d, void* stream) {
  if (!d || !d->obs || !d->act || !d->oldlp || !d->adv || !d->ret || !d->idx || !d->params || !d->grads || !d->adv_stats || !d->acc ||
      !d->workspace)
    return fail(MYO_E_ARG, "myo_ppo_mlp_step: null argument");
  const int B = d->B, O = d->O, A = d->A, OP = (O + 31) / 32 * 32;
  if (!mlp_shape_ok(B, O, A, d->hidden)) return fail(MYO_E_UNSUPPORTED, "myo_ppo_mlp_step: needs hidden = %d, obs <= %d, act <= %d, batch a multiple of %d", MLP_H, MLP_OPMAX, MLP_APM, 64 * MLP_SPLITK);
  const MlpWs w = mlp_carve((unsigned char*)d->workspace, B, OP, A, d->G);
  if ((long long)w.bytes > d->workspace_bytes) return fail(MYO_E_ARG, "myo_ppo_mlp_step: workspace too small (%zu bytes needed)", w.bytes);
  hipStream_t st = (hipStream_t)stream;
  static bool lds_set = false;
  if (!lds_set) {
    if (hipFuncSetAttribute((const void*)k_mlp_fwdbwd, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_FWDBWD_LDS) != hipSuccess)
      return fail(MYO_E_DEVICE, "myo_ppo_mlp_step: cannot reserve %d bytes of LDS", (int)MLP_FWDBWD_LDS);
    lds_set = true;
  }
  if (d->compute_adv_stats) hipLaunchKernelGGL(k_adv_moments, dim3(MLP_ADV_BLOCKS), dim3(256), 0, st, d->adv, (const long long*)d->idx, B, w.advpart);
  MlpPrepArgs pp;
  pp.p = d->params; pp.O = O; pp.OP = OP; pp.Ah[0] = A; pp.Ah[1] = 1;
  for (int k = 0; k < 2; ++k) {
    pp.off_W1[k] = d->off_W1[k]; pp.off_b1[k] = d->off_b1[k]; pp.off_W2[k] = d->off_W2[k]; pp.off_b2[k] = d->off_b2[k];
    pp.off_Wh[k] = d->off_Wh[k]; pp.off_bh[k] = d->off_bh[k];
  }
  pp.adv_part = w.advpart; pp.adv_stats = d->adv_stats; pp.adv_nb = d->compute_adv_stats ? MLP_ADV_BLOCKS : 0; pp.B = B;
  pp.W1p = w.W1p; pp.W2 = w.W2; pp.W2T = w.W2T; pp.Whp = w.Whp; pp.WhT = w.WhT; pp.bias = w.bias;
  hipLaunchKernelGGL(k_mlp_prep, dim3(256), dim3(256), 0, st, pp);
  MlpArgs a;
  a.obs = d->obs; a.act = d->act; a.oldlp = d->oldlp; a.adv = d->adv; a.ret = d->ret; a.idx = (const long long*)d->idx;
  a.log_std = d->params + d->off_log_std; a.adv_stats = d->adv_stats;
  a.W1p = w.W1p; a.W2 = w.W2; a.W2T = w.W2T; a.Whp = w.Whp; a.WhT = w.WhT; a.bias = w.bias;
  a.XT = w.XT; a.H1T = w.H1T; a.H2T = w.H2T; a.dH1T = w.dH1T; a.dH2T = w.dH2T; a.dOT = w.dOT; a.part = w.part;
  a.B = B; a.O = O; a.A = A; a.OP = OP; a.NB = B / MLP_BM; a.clip = d->clip; a.vf_coef = d->vf_coef;
  hipLaunchKernelGGL(k_mlp_fwdbwd, dim3(B / MLP_BM, 2), dim3(256), MLP_FWDBWD_LDS, st, a);
  MlpWgradArgs g;
  g.njobs = 0; g.B = B; g.rows_per_split = B / MLP_SPLITK; g.G = d->G; g.slab = w.slab;
  const size_t HB = (size_t)MLP_H * B;
  for (int net = 0; net < 2; ++net) {
    for (int m0 = 0; m0 < MLP_H; m0 += 128) {
      for (int n0 = 0; n0 < MLP_H; n0 += 128)
        g.job[g.njobs++] = MlpWgradJob{w.dH2T + net * HB, w.H1T + net * HB, MLP_H, MLP_H, MLP_H, m0, n0, d->off_W2[net], n0 == 0 ? d->off_b2[net] : -1};
      g.job[g.njobs++] = MlpWgradJob{w.dH1T + net * HB, w.XT, MLP_H, O, O, m0, 0, d->off_W1[net], d->off_b1[net]};
    }
    for (int n0 = 0; n0 < MLP_H; n0 += 128)
      g.job[g.njobs++] = MlpWgradJob{w.dOT + (size_t)net * 128 * B, w.H2T + net * HB, net == 0 ? A : 1, MLP_H, MLP_H, 0, n0, d->off_Wh[net], -1};
  }
  hipLaunchKernelGGL(k_mlp_wgrad, dim3(g.njobs * MLP_SPLITK), dim3(256), 0, st, g);
  if (d->sqnorm_part && d->adam_step) {
    MlpRfArgs rf;
    rf.slab = w.slab; rf.g = d->grads; rf.G = d->G; rf.splits = MLP_SPLITK; rf.part = w.part; rf.acc = d->acc; rf.NB = B / MLP_BM; rf.A = A;
    rf.ent_coef = d->ent_coef; rf.off_log_std = d->off_log_std; rf.off_bh0 = d->off_bh[0]; rf.off_bh1 = d->off_bh[1];
    rf.sq_part = d->sqnorm_part; rf.adam_step = d->adam_step;
    hipLaunchKernelGGL(k_mlp_reduce_finish, dim3(MLP_RF_BLOCKS + 2 * A + 3), dim3(256), 0, st, rf);
  } else {
    hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((d->G + 255) / 256)), dim3(256), 0, st, (const float*)w.slab, d->grads, d->G, MLP_SPLITK);
    hipLaunchKernelGGL(k_colmajor_finish, dim3(2 * A + 3), dim3(64), 0, st, (const float*)w.part, d->acc, B / MLP_BM, A, d->ent_coef,
                       d->grads + d->off_log_std, d->grads + d->off_bh[0], d->grads + d->off_bh[1]);
  }
  LAUNCH_CHECK(0)
  return MYO_OK;
}

struct MlpImg { unsigned short *W1p, *W2, *W2T, *Whp, *WhT; float* bias; size_t bytes; };
static MlpImg mlp_carve_images(unsigned char* base, int OP) {
  MlpImg w;
  size_t o = 0;
  auto take = [&](size_t n) { unsigned char* p = base ? base + o : nullptr; o += (n + 255) / 256 * 256; return p; };
  const size_t H = MLP_H;
  w.W1p = (unsigned short*)take(2 * H * OP * 2); w.W2 = (unsigned short*)take(2 * H * H * 2); w.W2T = (unsigned short*)take(2 * H * H * 2);
  w.Whp = (unsigned short*)take(2 * MLP_APM * H * 2); w.WhT = (unsigned short*)take(2 * H * MLP_AKP * 2);
  w.bias = (float*)take((4 * H + 2 * MLP_APM) * 4);
  w.bytes = o;
  return w;
}
static int mlp_rollout_ok(const myo_ppo_mlp_rollout_desc* d) {
  return d->hidden == MLP_H && d->O >= 1 && d->O <= MLP_OPMAX && d->A >= 1 && d->A <= MLP_APM && d->N >= MLP_BM && d->N % MLP_BM == 0;
}
extern "C" long long myo_ppo_mlp_rollout_workspace_bytes(int obs_dim, int act_dim, int hidden) {
  if (hidden != MLP_H || obs_dim < 1 || obs_dim > MLP_OPMAX || act_dim < 1 || act_dim > MLP_APM) return -1;
  return (long long)mlp_carve_images(nullptr, (obs_dim + 31) / 32 * 32).bytes;
}
extern "C" int myo_ppo_mlp_rollout_refresh(const myo_ppo_mlp_rollout_desc* d, void* stream) {
  if (!d || !d->params || !d->workspace) return fail(MYO_E_ARG, "myo_ppo_mlp_rollout_refresh: null argument");
  if (!mlp_rollout_ok(d)) return fail(MYO_E_UNSUPPORTED, "myo_ppo_mlp_rollout: needs hidden = %d, obs <= %d, act <= %d, envs a multiple of %d", MLP_H, MLP_OPMAX, MLP_APM, MLP_BM);
  const int OP = (d->O + 31) / 32 * 32;
  const MlpImg w = mlp_carve_images((unsigned char*)d->workspace, OP);
  if ((long long)w.bytes > d->workspace_bytes) return fail(MYO_E_ARG, "myo_ppo_mlp_rollout_refresh: workspace too small (%zu bytes needed)", w.bytes);
  MlpPrepArgs pp;
  pp.p = d->params; pp.O = d->O; pp.OP = OP; pp.Ah[0] = d->A; pp.Ah[1] = 1;
  for (int k = 0; k < 2; ++k) {
    pp.off_W1[k] = d->off_W1[k]; pp.off_b1[k] = d->off_b1[k]; pp.off_W2[k] = d->off_W2[k]; pp.off_b2[k] = d->off_b2[k];
    pp.off_Wh[k] = d->off_Wh[k]; pp.off_bh[k] = d->off_bh[k];
  }
  pp.adv_part = nullptr; pp.adv_stats = nullptr; pp.adv_nb = 0; pp.B = 0;
  pp.W1p = w.W1p; pp.W2 = w.W2; pp.W2T = w.W2T; pp.Whp = w.Whp; pp.WhT = w.WhT; pp.bias = w.bias;
  hipLaunchKernelGGL(k_mlp_prep, dim3(256), dim3(256), 0, (hipStream_t)stream, pp);
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_ppo_mlp_rollout(const myo_ppo_mlp_rollout_desc* d, void* stream) {
  if (!d || !d->obs || !d->params || !d->draw_counter || !d->t_idx || !d->act_buf || !d->val_buf || !d->logp_buf || !d->clipped || !d->workspace)
    return fail(MYO_E_ARG, "myo_ppo_mlp_rollout: null argument");
  if (!mlp_rollout_ok(d)) return fail(MYO_E_UNSUPPORTED, "myo_ppo_mlp_rollout: needs hidden = %d, obs <= %d, act <= %d, envs a multiple of %d", MLP_H, MLP_OPMAX, MLP_APM, MLP_BM);
  const int OP = (d->O + 31) / 32 * 32;
  const MlpImg w = mlp_carve_images((unsigned char*)d->workspace, OP);
  if ((long long)w.bytes > d->workspace_bytes) return fail(MYO_E_ARG, "myo_ppo_mlp_rollout: workspace too small (%zu bytes needed)", w.bytes);
  MlpPolicyArgs a;
  a.obs = d->obs; a.log_std = d->params + d->off_log_std; a.W1p = w.W1p; a.W2 = w.W2; a.Whp = w.Whp; a.bias = w.bias;
  a.obs_buf = d->obs_buf; a.act_buf = d->act_buf; a.val_buf = d->val_buf; a.logp_buf = d->logp_buf; a.clipped = d->clipped;
  a.t_idx = d->t_idx; a.draw_counter = (unsigned long long*)d->draw_counter; a.seed = d->seed;
  a.N = d->N; a.O = d->O; a.A = d->A; a.OP = OP; a.deterministic = d->deterministic;
  hipLaunchKernelGGL(k_mlp_policy, dim3(d->N / MLP_BM, 2), dim3(256), MLP_POLICY_LDS, (hipStream_t)stream, a);
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// ------------------------------------------------------------------------------------------ fused LSTM time step
#include "myo_lstm_step.h"
extern "C" int myo_lstm_step_supported(int H) { return (H == 32 || H == 64 || H == 128 || H == 256) ? 1 : 0; }
extern "C" int myo_lstm_step_fwd(const void* gx, long long gx_sg, long long gx_sr, const void* h_prev, const void* c_prev, const void* w_hh,
                                 const float* keep_next, int G, int N, int H, void* out_h, long long out_sg, void* hm_next, void* cm_next,
                                 void* c_new, void* ws, const float* c_prev32, float* cm_next32, void* stream) {
  if (!gx || !h_prev || (!c_prev && !c_prev32) || !w_hh || !out_h || !hm_next || (!cm_next && !cm_next32) || G <= 0 || N <= 0 || H <= 0 || (gx_sg & 3) || (gx_sr & 3) || (out_sg & 3))
    return fail(MYO_E_ARG, "myo_lstm_step_fwd: bad arguments");
  if (!myo_lstm_step_supported(H)) return fail(MYO_E_UNSUPPORTED, "myo_lstm_step_fwd: hidden size %d (32, 64, 128 or 256)", H);
  typedef const unsigned short* cu;
  typedef unsigned short* mu;
#define MYO_LSTM_FWD(HH)                                                                                                              \
  lstm_step_fwd_launch<HH>((cu)gx, gx_sg, gx_sr, (cu)h_prev, (cu)c_prev, (cu)w_hh, keep_next, G, N, (mu)out_h, out_sg, (mu)hm_next,   \
                           (mu)cm_next, (mu)c_new, (mu)ws, c_prev32, cm_next32, (hipStream_t)stream)
  switch (H) {
    case 32: MYO_LSTM_FWD(32); break;
    case 64: MYO_LSTM_FWD(64); break;
    case 128: MYO_LSTM_FWD(128); break;
    default: MYO_LSTM_FWD(256); break;
  }
#undef MYO_LSTM_FWD
  LAUNCH_CHECK(0)
  return MYO_OK;
}
extern "C" int myo_lstm_step_bwd(const void* dout, long long dout_sg, const void* dgates_next, const void* dcm_next, const void* w_hh_t,
                                 const float* keep_next, const void* c_prev, const void* c_new, const void* ws, int G, int N, int H,
                                 void* dgates, void* dc_prev, void* stream) {
  if (!c_prev || !c_new || !ws || !dgates || !dc_prev || (dgates_next && !w_hh_t) || G <= 0 || N <= 0 || H <= 0 || (dout_sg & 3))
    return fail(MYO_E_ARG, "myo_lstm_step_bwd: bad arguments");
  if (!myo_lstm_step_supported(H)) return fail(MYO_E_UNSUPPORTED, "myo_lstm_step_bwd: hidden size %d (32, 64, 128 or 256)", H);
  typedef const unsigned short* cu;
  typedef unsigned short* mu;
#define MYO_LSTM_BWD(HH)                                                                                                               \
  lstm_step_bwd_launch<HH>((cu)dout, dout_sg, (cu)dgates_next, (cu)dcm_next, (cu)w_hh_t, keep_next, (cu)c_prev, (cu)c_new, (cu)ws, G, N, \
                           (mu)dgates, (mu)dc_prev, (hipStream_t)stream)
  switch (H) {
    case 32: MYO_LSTM_BWD(32); break;
    case 64: MYO_LSTM_BWD(64); break;
    case 128: MYO_LSTM_BWD(128); break;
    default: MYO_LSTM_BWD(256); break;
  }
#undef MYO_LSTM_BWD
  LAUNCH_CHECK(0)
  return MYO_OK;
}

// whole sequences in one launch per direction (csrc/myo_lstm_seq.h)
#include "myo_lstm_seq.h"
extern "C" int myo_lstm_seq_supported(int H) { return (H == 128 || H == 256) ? 1 : 0; }
// row_split RS: a workgroup owns 16 / RS sequences (1, 2, or 4 at H = 256); the weight fragments and the tile-major arrays are laid out for it
static int lstm_seq_rs_ok(int H, int rs) { return rs == 1 || rs == 2 || (rs == 4 && H == 256); }
extern "C" int myo_lstm_seq_fwd(const void* gx, long long gx_st, long long gx_sg, long long gx_sr, void* hm, void* cm, const void* w_frag,
                                const float* keep, int G, int N, int H, int T, int row_split, void* out_h, long long out_sg, long long out_st,
                                void* c_new, void* ws, const float* c0_32, void* stream) {
  if (!gx || !hm || !cm || !w_frag || !keep || !out_h || !c_new || !ws || G <= 0 || N <= 0 || H <= 0 || T <= 0 || (gx_st & 7) || (gx_sg & 7) ||
      (gx_sr & 7) || (out_sg & 7) || (out_st & 7) || (N & 15))
    return fail(MYO_E_ARG, "myo_lstm_seq_fwd: bad arguments (N must be a multiple of 16, strides of 8)");
  if (!myo_lstm_seq_supported(H) || !lstm_seq_rs_ok(H, row_split))
    return fail(MYO_E_UNSUPPORTED, "myo_lstm_seq_fwd: hidden size %d (128 or 256) with row split %d (1, 2; 4 at 256)", H, row_split);
  typedef const unsigned short* cu;
  typedef unsigned short* mu;
#define MYO_SEQ_FWD(HH, RR) lstm_seq_fwd_launch<HH, RR>((cu)gx, gx_st, gx_sg, gx_sr, (mu)hm, (mu)cm, (cu)w_frag, keep, G, N, T, (mu)out_h, out_sg, out_st, \
                                                        (mu)c_new, (mu)ws, c0_32, (hipStream_t)stream)
  const int rc = H == 128 ? (row_split == 1 ? MYO_SEQ_FWD(128, 1) : MYO_SEQ_FWD(128, 2))
                          : (row_split == 1 ? MYO_SEQ_FWD(256, 1) : row_split == 2 ? MYO_SEQ_FWD(256, 2) : MYO_SEQ_FWD(256, 4));
#undef MYO_SEQ_FWD
  return rc ? fail(MYO_E_DEVICE, "myo_lstm_seq_fwd: launch failed") : MYO_OK;
}
extern "C" int myo_lstm_seq_bwd(const void* dout, long long dout_sg, long long dout_st, const void* wt_frag, const float* keep, const void* cm,
                                const void* c_new, const void* ws, int G, int N, int H, int T, int row_split, void* dgates, void* stream) {
  if (!dout || !wt_frag || !keep || !cm || !c_new || !ws || !dgates || G <= 0 || N <= 0 || H <= 0 || T <= 0 || (dout_sg & 7) || (dout_st & 7) || (N & 15))
    return fail(MYO_E_ARG, "myo_lstm_seq_bwd: bad arguments (N must be a multiple of 16, strides of 8)");
  if (!myo_lstm_seq_supported(H) || !lstm_seq_rs_ok(H, row_split))
    return fail(MYO_E_UNSUPPORTED, "myo_lstm_seq_bwd: hidden size %d (128 or 256) with row split %d (1, 2; 4 at 256)", H, row_split);
  typedef const unsigned short* cu;
#define MYO_SEQ_BWD(HH, RR) lstm_seq_bwd_launch<HH, RR>((cu)dout, dout_sg, dout_st, (cu)wt_frag, keep, (cu)cm, (cu)c_new, (cu)ws, G, N, T, \
                                                        (unsigned short*)dgates, (hipStream_t)stream)
  const int rc = H == 128 ? (row_split == 1 ? MYO_SEQ_BWD(128, 1) : MYO_SEQ_BWD(128, 2))
                          : (row_split == 1 ? MYO_SEQ_BWD(256, 1) : row_split == 2 ? MYO_SEQ_BWD(256, 2) : MYO_SEQ_BWD(256, 4));
#undef MYO_SEQ_BWD
  return rc ? fail(MYO_E_DEVICE, "myo_lstm_seq_bwd: launch failed") : MYO_OK;
}
