// wave.h — "one environment per 64-lane wavefront" programming layer.
//
// The stepper is written as a sequence of PHASES.  Inside a phase every lane works on its own
// item (a body, a tendon, a dof, a constraint row …) and only READS what earlier phases wrote to
// the per-env scratch (LDS); a SYNC() separates phases.  Values that all lanes agree on
// ("uniform": counts, loop bounds, step sizes) live in the scratch or are produced by the
// WAVE_* reductions below.
//
// gfx950 build (hipcc): a workgroup is exactly one wavefront (64 threads), `lane` is
// threadIdx.x, PHASE expands to nothing, SYNC() is a wavefront-scope FENCE — no instruction: the
// wavefront's LDS operations execute in program order — and SYNC_G() the drain that global-memory
// hand-offs between lanes need (see below), reductions use DPP cross-lane adds, compaction uses
// 64-bit ballots.
//
// MYO_EMU build (g++, tests only): PHASE expands to `for (lane = 0..63)`, i.e. the lanes of a
// phase run one after another on the CPU.  This exists so the kernel SOURCE can be debugged
// and run under AddressSanitizer/UBSan without a GPU (GPU ASan is unavailable on the pool).
// It is compiled only by tests/emu/ into a separate library; libmyobatch.so contains no CPU
// path.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef MYO_EMU
#define DEV static inline
#define WAVE_FN
#define WAVE_FN_K
#define PHASE for (int lane = 0; lane < 64; ++lane)
#define SYNC() ((void)0)
#define SYNC_G() ((void)0)
#define GPTR(T) T*
#define MYO_PIN(x) ((void)0)
#define LANE_VAR(T, name) T name[64]
#define LV(name) name[lane]
// sum over i in [0,n) of expr(i); result uniform
#define WAVE_SUM_N(T, out, n, i, expr) \
  T out = 0;                           \
  for (int i = 0; i < (n); ++i) { out += (expr); }
#define WAVE_SUM3_N(T, o1, o2, o3, n, i, ...)   \
  T o1 = 0, o2 = 0, o3 = 0;                      \
  for (int i = 0; i < (n); ++i) {                \
    T _e1 = 0, _e2 = 0, _e3 = 0;                 \
    __VA_ARGS__;                                 \
    o1 += _e1; o2 += _e2; o3 += _e3;             \
  }
// sums over the 64 lanes of two per-lane expressions written with LV() (lane variables)
#define WAVE_SUM2_LANES(T, o1, o2, ...)          \
  T o1 = 0, o2 = 0;                              \
  for (int lane = 0; lane < 64; ++lane) {        \
    T _e1 = 0, _e2 = 0;                          \
    __VA_ARGS__;                                 \
    o1 += _e1; o2 += _e2;                        \
  }
// exclusive prefix over lanes of a per-lane count in {0,1,2}; pre = scratch int[64]
#define WAVE_EXSCAN(cnt_expr, pre, total)            \
  {                                                  \
    int _t = 0;                                      \
    for (int lane = 0; lane < 64; ++lane) {          \
      (pre)[lane] = _t;                              \
      _t += (cnt_expr);                              \
    }                                                \
    (total) = _t;                                    \
  }
// the same for counts up to 6 (contact slots: two contacts of up to three slots per lane)
#define WAVE_EXSCAN6(cnt_expr, pre, total) WAVE_EXSCAN(cnt_expr, pre, total)
#define UNI(x) (x)
// LDS accumulation by several lanes of a phase into one slot (the emulation runs the lanes one after the other)
template <typename T> static inline void lds_add(T* p, T v) { *p += v; }
static inline void myo_count(int* p) { *p += 1; }
static inline void myo_max(int* p, int v) { if (v > *p) *p = v; }
// a store another workgroup of the launch reads after its agent acquire (st_pub of the device build): the emulation has one memory
static inline void st_pub(double* p, double v, int wt) { (void)wt; *p = v; }
#define MYO_WAVE_SLOTS_EMU 0
static inline unsigned myo_wave_slot(int env) { return (unsigned)env; }      /* the emulation keeps one workspace per env */
static inline int myo_ws_acquire(int* owner, int env, int* health) { (void)owner; (void)health; return env; }
static inline void myo_ws_release(int* owner, int idx) { (void)owner; (void)idx; }
static inline int myo_popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int myo_ffsll(unsigned long long x) { return __builtin_ctzll(x); }
#else
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#define WAVE_FN const int lane = threadIdx.x; (void)lane;
// ... of the INLINED functions that make up the kernel body (the step, the substep, the Newton loop), fp64 stepper: the lane index behind
// an empty asm, so that what is derived from it (LDS addresses, 64-bit global addresses) is computed where it is used.  Without it the
// compiler hoists ~100 lane-derived addresses to the top of the kernel and keeps them live across every call: 28-71 VGPR spills in the
// kernel frame, reloaded inside the Newton loop (k_step<double>: .vgpr_spill_count 28 -> 0, the die's 48-slot stepper 71 -> 0 and
// -4.7 % kernel time; the fp32 stepper is 1.7 % slower with it, its addresses are cheaper to keep than to recompute: not pinned).
template <bool PIN> __device__ __forceinline__ int myo_lane() { int l = threadIdx.x; if constexpr (PIN) asm volatile("" : "+v"(l)); return l; }
#define WAVE_FN_K const int lane = myo_lane<sizeof(T) == 8>(); (void)lane;
// A store of bytes that ANOTHER workgroup of the same launch reads after its agent-scope acquire (the parts of an env step, k_step in
// myobatch.hip).  wt != 0: a relaxed agent-scope store = `global_store_dwordx2 ... sc1`, written through the XCD's L2, so that the
// publishing lane needs no agent release fence (`buffer_wbl2 sc1` writes back EVERY dirty line of the XCD's L2, the other envs'
// workspace lines included) — only its `s_waitcnt vmcnt(0)` before the flag.  wt == 0: a plain store, published by the release fence.
__device__ __forceinline__ void st_pub(double* p, double v, int wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
#define PHASE
// SYNC(): the phase boundary.  A workgroup is ONE wavefront: its LDS instructions execute in program order (a ds_write of one lane
// followed by a ds_read of another needs no wait between them), cross-lane register traffic (DPP, readlane) is ordered by the
// instruction stream, and what a later phase reads from LDS is waited for by the compiler's own s_waitcnt on the destination
// register.  So the boundary is a WAVEFRONT-scope fence: it keeps the compiler from moving memory accesses across it and emits no
// instruction.  __syncthreads() (rounds 1-5) lowered to `s_waitcnt vmcnt(0) lgkmcnt(0)` at every one of the ~400 boundaries of a
// substep: each drained the LDS stores of the phase just ended (and every outstanding table prefetch / workspace store) before the
// next phase could issue its first load.
// SYNC_G(): the boundary after a phase whose GLOBAL-memory stores (the wave slot's workspace, TaskDev::ctrl_ws) are read by OTHER
// lanes later: drains the stores (what SYNC() was).  Lanes that only re-read what they stored themselves need neither.
// -DMYO_SYNC_FULL restores the full drain everywhere (A/B).
#ifdef MYO_SYNC_FULL
#define SYNC() __syncthreads()
#else
#define SYNC() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); asm volatile("" ::: "memory"); }
#endif
#define SYNC_G() __syncthreads()
// a value that must exist in a vector register HERE: keeps a table load where it was written (the compiler otherwise sinks a load into
// the one branch that uses its result — and waits for it there, one memory round trip per branch)
#define MYO_PIN(x) asm volatile("" : "+v"(x))
// a pointer into GLOBAL memory kept in LDS or in the task constants: typed as such, so that its accesses are global_load / global_store
// and not flat_* (a flat access is not ordered with the wave's ds_* instructions when it lands in LDS, and it counts on both wait counters)
#define GPTR(T) __attribute__((address_space(1))) T*
__device__ __forceinline__ void st_pub(GPTR(double) p, double v, int wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
#define LANE_VAR(T, name) T name
#define LV(name) name
// Wave-wide sum in ~11 VALU instructions: DPP butterflies inside each 16-lane row
// (quad_perm xor1, xor2, row_half_mirror, row_mirror), then the four row totals are read with
// v_readlane and added as scalars.  (A __shfl_xor chain lowers to 6 dependent ds_bpermute round
// trips through the LDS crossbar, ~10x slower; the solver's line search does dozens per substep.)
__device__ __forceinline__ float myo_dpp_f(float v, const int ctrl_sel) {
  const int i = __float_as_int(v);
  int r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_mov_dpp(i, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
    default: r = __builtin_amdgcn_mov_dpp(i, 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return __int_as_float(r);
}
__device__ __forceinline__ double myo_dpp_d(double v, const int ctrl_sel) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T> __device__ __forceinline__ T myo_wave_sum(T v);
template <> __device__ __forceinline__ float myo_wave_sum<float>(float v) {
  v += myo_dpp_f(v, 0); v += myo_dpp_f(v, 1); v += myo_dpp_f(v, 2); v += myo_dpp_f(v, 3);
  const int i = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return (r0 + r1) + (r2 + r3);
}
template <> __device__ __forceinline__ double myo_wave_sum<double>(double v) {
  v += myo_dpp_d(v, 0); v += myo_dpp_d(v, 1); v += myo_dpp_d(v, 2); v += myo_dpp_d(v, 3);
  const long long b = __double_as_longlong(v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  double r[4];
  for (int k = 0; k < 4; ++k) {
    const int l = __builtin_amdgcn_readlane(lo, 16 * k), h = __builtin_amdgcn_readlane(hi, 16 * k);
    r[k] = __longlong_as_double(((long long)h << 32) | (unsigned int)l);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}
template <> __device__ __forceinline__ int myo_wave_sum<int>(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
#define WAVE_SUM_N(T, out, n, i, expr)                       \
  T out;                                                     \
  {                                                          \
    T _a = 0;                                                \
    for (int i = lane; i < (n); i += 64) { _a += (expr); }   \
    out = myo_wave_sum<T>(_a);                               \
  }
#define WAVE_SUM3_N(T, o1, o2, o3, n, i, ...)               \
  T o1, o2, o3;                                              \
  {                                                          \
    T _a1 = 0, _a2 = 0, _a3 = 0;                             \
    for (int i = lane; i < (n); i += 64) {                   \
      T _e1 = 0, _e2 = 0, _e3 = 0;                           \
      __VA_ARGS__;                                           \
      _a1 += _e1; _a2 += _e2; _a3 += _e3;                    \
    }                                                        \
    o1 = myo_wave_sum<T>(_a1);                               \
    o2 = myo_wave_sum<T>(_a2);                               \
    o3 = myo_wave_sum<T>(_a3);                               \
  }
#define WAVE_SUM2_LANES(T, o1, o2, ...)                      \
  T o1, o2;                                                  \
  {                                                          \
    T _e1 = 0, _e2 = 0;                                      \
    __VA_ARGS__;                                             \
    o1 = myo_wave_sum<T>(_e1);                               \
    o2 = myo_wave_sum<T>(_e2);                               \
  }
#define WAVE_EXSCAN(cnt_expr, pre, total)                                          \
  {                                                                                \
    int _c = (cnt_expr);                                                           \
    unsigned long long _m1 = __ballot(_c >= 1), _m2 = __ballot(_c >= 2);          \
    unsigned long long _below = (1ull << lane) - 1ull;                             \
    (pre)[lane] = __popcll(_m1 & _below) + __popcll(_m2 & _below);                 \
    (total) = __popcll(_m1) + __popcll(_m2);                                       \
  }                                                                                \
  SYNC();
#define WAVE_EXSCAN6(cnt_expr, pre, total)                                         \
  {                                                                                \
    const int _c = (cnt_expr);                                                     \
    const unsigned long long _below = (1ull << lane) - 1ull;                       \
    int _p = 0, _t = 0;                                                            \
    _Pragma("unroll") for (int _k = 1; _k <= 6; ++_k) {                            \
      const unsigned long long _m = __ballot(_c >= _k);                            \
      _p += __popcll(_m & _below); _t += __popcll(_m);                             \
    }                                                                              \
    (pre)[lane] = _p; (total) = _t;                                                \
  }                                                                                \
  SYNC();
// LDS accumulation by several lanes of a phase into one slot: ds_add_f32 / ds_add_f64 without return value.  Lanes of one
// instruction that hit the same slot are served in a fixed order by the LDS unit, so the sum is reproducible.
template <typename T> __device__ __forceinline__ void lds_add(T* p, T v) {
  typedef __attribute__((address_space(3))) T* lds_p;
#ifdef MYO_LDS_ADD_SERIAL
  // diagnostic build: the adds of a wave one lane after the other, in lane order (is the order of same-address ds_add_f64 of one instruction fixed?)
  const unsigned long long act = __ballot(1);
  for (unsigned long long m = act; m; m &= m - 1) {
    const int l = __ffsll((long long)m) - 1;
    if ((int)threadIdx.x == l) *(lds_p)p += v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
#else
  (void)__hip_atomic_fetch_add((lds_p)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ void myo_count(int* p) { atomicAdd(p, 1); }     // event counter in global memory
__device__ __forceinline__ void myo_max(int* p, int v) { atomicMax(p, v); }
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
// The hardware wave slot this wavefront occupies, as an index: (XCC, SE, SH, CU, SIMD, wave buffer) from HW_REG_XCC_ID [3:0] and
// HW_REG_HW_ID (gfx9 / gfx94x / gfx950 layout: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]).
// Two wavefronts that are resident at the same time never share it, and a slot belongs to one XCD (checked on the device by
// myo_debug_wave_slots, tests/test_step_parts.py, -m gpu) — but a wavefront may be moved to another slot mid-kernel: see myo_ws_acquire.
#define MYO_WAVE_SLOTS (8 * 16384)
__device__ __forceinline__ unsigned myo_wave_slot(int) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  return ((xcc & 7u) << 14) | (((hw >> 8) & 0xffu) << 6) | (hw & 0x3fu);
}
// A WORKSPACE block for this workgroup: the block the hardware slot it sits in prefers (a hash of the slot code inside the XCD's range
// of MYO_WS_SLOTS / 8 blocks — so a slot keeps meeting the same, L2-resident block), taken with a compare-and-swap on its owner flag
// and given back when the workgroup leaves (myo_ws_release); if the flag is taken, the next free block of the range.  Ownership, not
// the slot, is what makes the block private: a wavefront does NOT keep its hardware slot for life — the scheduler saves and restores
// waves mid-kernel (measured on MI355X: of 30 k one-substep launches x 4096 workgroups, 1,517 workgroups ended in another slot than
// they started in, and 61 found the slot they arrived in still in use) — and round 6's first form, block = f(slot) without a flag, let a
// restored wave and the slot's next tenant write one block: one k_step launch in ~5,000 left a quarter of its envs in other last
// bits than the same launch of an identical run (tools/dev/soak_*.py).  An exhausted range (never: 2048 blocks for
// the <= 512 workgroups an XCD holds) is counted as a protocol error (health[0]).
#define MYO_WS_SLOTS 16384
#define MYO_WS_PER_XCD (MYO_WS_SLOTS / 8)
__device__ __forceinline__ int myo_ws_acquire(int* owner, int, int* health) {
  const unsigned code = myo_wave_slot(0);
  const int base = (int)((code >> 14) & 7u) * MYO_WS_PER_XCD;
  const unsigned pref = ((code & 0x3fffu) * 2654435761u) >> 21;          // 11 bits
  for (int k = 0; k < MYO_WS_PER_XCD; ++k) {
    const int idx = base + (int)((pref + (unsigned)k) & (MYO_WS_PER_XCD - 1));
    if (atomicCAS(owner + idx, 0, 1) == 0) return idx;
  }
  if (health) atomicAdd(health, 1);
  return base + (int)pref;
}
__device__ __forceinline__ void myo_ws_release(int* owner, int idx) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's stores into the block are in L2 before the next owner's
  (void)atomicExch(owner + idx, 0);
}
__device__ __forceinline__ int myo_popcll(unsigned long long x) { return __popcll(x); }
__device__ __forceinline__ int myo_ffsll(unsigned long long x) { return __ffsll((long long)x) - 1; }
#endif
