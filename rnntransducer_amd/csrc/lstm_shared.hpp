// Shared by lstm.hip and lstm5.hip: kernel argument block, inter-workgroup protocol helpers and the grouped-launch helper of the
// persistent recurrences (see the header comment of lstm.hip).  Everything here is per-translation-unit (anonymous namespace).
#pragma once
#include "common.hpp"

#include <stdlib.h>

#include <type_traits>

namespace rnnt {

struct LstmK {
  int T, B, H, D, Hs, NC, Bp, LDW;
  const int* lens;
  float* gates;   // (T,B,D,4H) gate-adjacent layout
  float* cst;     // (D,T,H/4,B,4)
  float* y;       // (T,B,D,H)
  float* ydrop;   // or nullptr
  float keep_scale;
  unsigned drop_thresh;
  unsigned long long seed;
  const float* w_hh[2];
  float* hx;       // fwd: [2][D][H/4][Bp][4]   bwd: [2][D][H][Bp][4]
  unsigned* flags; // [D][NC]
  unsigned* status;
  const float* dy; // bwd: (T,B,D,H)
  int G, Bg, Kp;   // v2: batch groups per direction, rows per group, K padded to a multiple of 64
  unsigned long long* dbg;  // diagnostic builds only (RNNT_LSTM_DBG): per-workgroup phase cycle sums, else nullptr
  int cell;        // RNNT_CELL_* (v2 kernels; v1 is LSTM only)
  const float* b_hh[2];  // GRU: the hidden-side bias of the n gate stays inside r * (W_hn h + b_hn)
  float* aux;      // GRU backward: hidden-side gate gradients (T,B,D,4H) for dW_hh / db_hh
  unsigned* xcc;   // v2: [D*G][NC] XCC id + 1 of every member, published once at kernel start (zeroed per launch)
  int allow_local; // v2: permit the L2-local exchange when a group is verified to sit on one XCD
  float* dbp;      // v4 backward: (D*G*NBR, 4H) time sums of the input-side dG per exchange row, or nullptr
  float* dbhp;     // same for the hidden-side dG (GRU)
  int NGL;         // v2+: launch stride of the group index (>= D*G): gid = blockIdx % NGL, blocks with gid >= D*G exit at once
  unsigned* colmax;    // v5 backward (or nullptr): per gate column (D*4H) running maximum of |dG| over all frames and rows (fp32 bit
  unsigned* colmax_h;  // patterns, zeroed by the host), input side / hidden side (GRU) — the column scales of the half-pair dG^T planes
  unsigned* rowmax;    // v5 backward, LSTM / Elman (or nullptr): per frame row (T*B) maximum of |dG| over all gate columns of both
                       // directions (zeroed by the host) — the row scales of the half-pair dG planes
  int pause;       // v5: s_sleep units (64 cycles) between a step's publication and its first poll round: low byte forward / backward
                   // consumers of early blocks, second byte backward consumers of late blocks (RNNT_LSTM_FWD_PAUSE, RNNT_LSTM_BWD_PAUSE=e,l)
  int gbound;      // v5: 1 = every sync group runs max(lens of its rows) steps instead of T (ragged batches, rnnt_lstm_desc.row_idx)
  int hw_math;     // v2: v_exp_f32 / v_rcp_f32 cell math (default; measured whole-model loss delta identical to the ocml expf +
                   // IEEE-division form, which RNNT_LSTM_EXACT_MATH=1 selects)
};

struct Plan2 {
  int HS, NC, G, Bg, BQ, Kp;
  int MB = 4;  // register-form kernels: 16-gate-column blocks per workgroup (HS = 4*MB)
  size_t lds_fwd, lds_bwd;
};

// v5 recurrences (lstm5.hip)
bool lstm5_supported(int T, int B, int H, int D, int cell);
int lstm5_fwd_launch(const LstmK& k, const Plan2& pl, int cell, hipStream_t s);
int lstm5_bwd_launch(const LstmK& k, const Plan2& pl, int cell, hipStream_t s);

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned long long SPIN_LIMIT_TICKS = 400000000ull;  // 4 s of the 100 MHz s_memrealtime clock
constexpr int RSRC_FLAGS = 0x00027000;                           // gfx9 raw buffer: 32-bit elements
constexpr int AUX_SC1 = 16;



// Per-phase cycle stamps of the step loops: compiled in only with -DRNNT_LSTM_DBG_STAMPS=1 (tools/lstm_phase_probe.py builds that
// variant: `python -m rnntransducer_amd.csrc.build --variant dbg --only=lstm.hip,lstm5.hip -DRNNT_LSTM_DBG_STAMPS=1`).  Even with
// p.dbg == nullptr the six `if (p.dbg && tid == 0)` tests per step cost the latency-bound loops 4.5 % (forward) / 3.5 % (backward) of
// their time at config 2 (paired runs: 6.08 -> 5.81 and 7.43 -> 7.17 ms per training step): the step is instruction-issue-bound in the
// gathering waves, not only exchange-latency-bound.
#ifdef RNNT_LSTM_DBG_STAMPS
#define DBG_STAMP(i) do { if (p.dbg && tid == 0) { const unsigned long long now_ = clock64(); dsum[i] += now_ - dlast; dlast = now_; } } while (0)
#else
#define DBG_STAMP(i) do { } while (0)
#endif

// counter-based dropout mask: murmur3-style 32-bit finaliser of (seed, element index) -- a dozen 32-bit ops, the
// forward and backward kernels regenerate the same mask from the same (seed, index)
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long idx) {
  unsigned h = (unsigned)idx ^ (unsigned)seed;
  h ^= ((unsigned)(idx >> 32) + (unsigned)(seed >> 32)) * 0x27d4eb2fu;
  h *= 0x9E3779B1u;
  h ^= h >> 15;
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float sig_sel(float x, int hw) { return hw ? sigmoid_hw(x) : sigmoidf_(x); }
__device__ __forceinline__ float tanh_sel(float x, int hw) { return hw ? tanh_hw(x) : tanh_e(x); }
// value of lane (l + n) within the same 16-lane row (n = 4, 8, 12): one DPP move, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float row_shl(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// Wave 0 polls this direction's step flags until all are >= need; everyone then joins a barrier.
// Returns false (uniformly) if the wait was abandoned (timeout or another workgroup raised the status word).
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int nflags, unsigned need, unsigned* status,
                                           int* abort_lds) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int bad = 0;
    unsigned spins = 0;
    while (true) {
      bool ok = true;
      for (int i = lane; i < nflags; i += 64)
        ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need);
      if (__all(ok)) break;
      if ((++spins & 63u) == 0u) {
        unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st != 0u || wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
          if (lane == 0 && st == 0u) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = 1;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) *abort_lds = bad;
  }
  __syncthreads();
  return *abort_lds == 0;
}

// Every storing wave drains its (write-through) stores, the workgroup meets, ONE lane publishes the flag.
__device__ __forceinline__ void publish_flag(unsigned* flag, unsigned value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- XCD-local exchange (speed only; correctness never depends on placement) -------------------------------------
// All CUs of one XCD share one L2.  When every member of a sync group is VERIFIED at run time (HW_REG_XCC_ID, exchanged
// once through the placement-independent sc1 protocol) to sit on the same XCD, the group switches its per-step
// exchange to plain stores (the line stays in that L2) + sc1 loads (bypass L1, hit L2): measured 1.96 vs 2.74 us per
// round in tools/sync_probe.hip.  A group that spans XCDs keeps the write-through protocol.  The decision is uniform
// across a group because every member evaluates the same published table.
__device__ __forceinline__ bool group_is_xcd_local(unsigned* xtab, int nmem, int me, unsigned* status, int* lds_flag) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc = (xcc & 0xfu) + 1u;
  if (threadIdx.x == 0) __hip_atomic_store(xtab + me, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!wait_flags(xtab, nmem, 1u, status, lds_flag)) return false;
  if (threadIdx.x < 64) {
    bool same = true;
    for (int i = threadIdx.x; i < nmem; i += 64)
      same = same && (__hip_atomic_load(xtab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc);
    same = __all(same);
    if (threadIdx.x == 0) *lds_flag = same ? 2 : 0;
  }
  __syncthreads();
  const bool local = *lds_flag == 2;
  __syncthreads();
  if (threadIdx.x == 0) *lds_flag = 0;
  __syncthreads();
  return local;
}

template <bool LOCAL>
__device__ __forceinline__ void exchange_store(i32x4 v, __amdgpu_buffer_rsrc_t r, int off) {
  if constexpr (LOCAL) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX_SC1);
}
template <bool LOCAL>
__device__ __forceinline__ void publish_flag2(unsigned* flag, unsigned value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if constexpr (LOCAL) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


inline int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}




template <typename K>
int launch_persistent2(K kernel, const LstmK& k_in, const Plan2& pl, size_t lds, hipStream_t s, const char* what, int threads = 256) {
  if (lds > 64 * 1024)
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  RNNT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
  const int cus = device_cus();
  LstmK k = k_in;
  const int NG = k.D * pl.G;
  // Launch stride of the group index: 8 (= XCDs) when every group fits one XCD, so that the round-robin block -> XCD placement
  // puts all members of a group on one XCD (L2-local exchange); blocks of the unused group slots exit at once.  A group of
  // more than 32 members fits an XCD only if this kernel's resources let two workgroups share a CU (register-form kernels).
  const int fit = (per_cu < 2 ? per_cu : 2) * (cus / 8);
  k.NGL = (NG <= 8 && pl.NC <= fit && !getenv("RNNT_LSTM_NO_XCD_STRIDE")) ? 8 : NG;
  const int active = NG * pl.NC;   // workgroups that take part (must be co-resident)
  const int grid = k.NGL * pl.NC;  // launched
  if (per_cu < 1 || active > cus * (k.NGL == 8 && pl.NC > cus / 8 ? 2 : 1)) {
    set_error("%s: %d workgroups cannot be co-resident (%d CUs x %d per CU)", what, active, cus, per_cu);
    return RNNT_ERR_UNSUPPORTED;
  }
  {
    const double per_tb = (k.dy ? (8.0 + 2.0 + 1.0) : (8.0 + 1.0 + 1.0)) * k.H * 4.0;
    ProfScope prof(k.dy ? RNNT_K_LSTM_BWD : RNNT_K_LSTM_FWD, per_tb * k.T * k.B * k.D + 4.0 * 4.0 * k.H * k.H * k.D, s);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, s, k);
  }
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

}  // namespace
}  // namespace rnnt
