// Shared helpers for librnnt_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "rnnt_hip.h"

namespace rnnt {

void set_error(const char* fmt, ...);

#define RNNT_CHECK_ARG(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      ::rnnt::set_error(__VA_ARGS__);  \
      return RNNT_ERR_INVALID;         \
    }                                  \
  } while (0)

#define RNNT_CHECK_HIP(expr)                                                          \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      ::rnnt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RNNT_ERR_LAUNCH;                                                         \
    }                                                                                 \
  } while (0)

#define RNNT_CHECK_LAUNCH() RNNT_CHECK_HIP(hipGetLastError())

// Opt-in profiler scope (api.hip): when rnnt_hip_prof_enable(1) was called, brackets the enclosed launches with
// HIP events on `s`; `work` is the algorithmic FLOPs (GEMM) or bytes (everything else) of the launch.
struct ProfScope {
  ProfScope(int kind, double work, hipStream_t s);
  ~ProfScope();
  int kind_;
  double work_;
  hipStream_t stream_;
  void* start_;
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// half-pair (hp) operands and the f16-MFMA GEMM on them (gemm_hp.hip); used by lstm.hip for its big products
size_t hp_plane_bytes(int64_t rows, int64_t K);
int hp_colmax(const float* x, int64_t rows, int64_t C, int64_t ld, uint32_t* amax, hipStream_t s);   // amax[C] of the columns
// Ragged batches: `rowidx` / `kidx` (device tables, or nullptr) list the VALID rows of a padded (T*B, .) tensor in time-major order.
//   hp_split: only the `rows` listed rows are converted, in place (plane row and amax entry rowidx[i]);
//   hp_split_t: contraction index k reads source row kidx[k] (+ shift), K = number of listed rows (planes packed along k);
//   hp_split_both: M = number of listed rows; row-major half-lines in place, transposed planes packed along k;
//   hp_gemm: row m of the product fetches plane row a_rowidx[m] of A (a_plane_rows = rows of those planes) and stores output row
//            c_rowidx[m] — the fetch of an operand tile gathers, no packed copy of a row-major operand is ever made.
int hp_split(const float* x, int64_t rows, int64_t K, int64_t ld, uint32_t* amax, void* planes, hipStream_t s, const int* rowidx = nullptr);  // writes amax[rows]
int hp_split_t(const float* x, int64_t R, int64_t K, int64_t ld, int64_t Ksrc, int64_t shift, const uint32_t* amax, void* planes,
               hipStream_t s, const int* kidx = nullptr);
// both orientations in one pass; rowmax[M] and colmax[C] given
int hp_split_both(const float* x, int64_t M, int64_t C, int64_t ld, const uint32_t* rowmax, const uint32_t* colmax, void* planes_rm,
                  void* planes_t, hipStream_t s, const int* rowidx = nullptr);
size_t hp_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int hp_gemm(const void* A, const uint32_t* a_amax, const void* B, const uint32_t* b_amax, int64_t M, int64_t N, int64_t K, float* C,
            int64_t c_div, int64_t c_so, int64_t c_si, const float* bias, unsigned flags, void* workspace, size_t workspace_bytes,
            hipStream_t s, const int* a_rowidx = nullptr, int64_t a_plane_rows = 0, const int* c_rowidx = nullptr);
// several NT products in ONE queue-driven launch (workgroups on the XCDs of `xcd_skip` leave at once): the weight-gradient
// products that run beside the next layer's recurrence.  `counter`: 16 device words, zeroed here: [0..7] one queue per XCD, [8] units completed, [9] set to 1 by the check
// kernel behind the launch when [8] != the number of units (a device that exposes fewer XCDs than `xcd_skip` assumes: every workgroup
// left and nothing was computed) — and then `status` (optional sticky device word, the recurrences' rnnt_lstm_desc.status) is raised to 2.
struct HpProblem {
  const void* A; const uint32_t* a_amax; const void* B; const uint32_t* b_amax;
  int64_t M, N, K;
  float* C; int64_t ldc;
  unsigned flags;
};
constexpr int HP_GROUP_MAX = 4;
size_t hp_gemm_grouped_workspace_bytes(const int64_t* MN, int n);   // MN[i] = M_i * N_i
int hp_gemm_grouped(const HpProblem* pr, int n, unsigned xcd_skip, unsigned* counter, void* workspace, size_t workspace_bytes,
                    hipStream_t s, unsigned* status = nullptr);
constexpr size_t HPQ_HEADER_BYTES = 256;   // room for `counter` in front of the slabs when both come out of one workspace
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// gelu_tanh and its derivative (torch.nn.GELU(approximate="tanh"), networks/transducer.py:38)
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float dgelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float x2 = x * x;
  float u = k0 * (x + k1 * x * x2);
  float th = tanhf(u);
  float du = k0 * (1.0f + 3.0f * k1 * x2);
  return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * du;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through one exp: |abs err| ~ 1e-7, saturates to +-1 (exp -> inf / 0); much shorter than ocml tanhf on the
// per-timestep critical path of the recurrences
__device__ __forceinline__ float tanh_e(float x) { return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f); }
// hardware-transcendental forms for the per-timestep critical path of the persistent recurrences, where ONE wave
// issues one VALU instruction per 4 cycles and ocml expf + IEEE division cost ~100 instructions per cell:
// v_exp_f32 / v_rcp_f32 are 1 ulp each -> |abs err| ~ 2e-7 on values in [-1, 1]; saturate correctly at +-inf.
// (Not __frcp_rn: the correctly rounded reciprocal is a 10-instruction refinement per call, and five of them sat on the serial
// chain of every forward recurrence step — c2: 8.0 -> 7.65 ms of forward recurrences per training step.)
__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_hw(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Exact fp32 -> bf16 piece split (gemm.hip, lstm.hip): x = x0 + x1 + x2 with x0 = x & 0xffff0000, r = x - x0 (exact),
// x1 = r & 0xffff0000, x2 = r - x1 (<= 8 significant bits left): every piece is a bf16 and piece products are exact in an
// fp32 accumulator, so sum_{i+j<=2} a_i b_j reproduces the fp32 product to O(2^-24).
// (even-k, odd-k) pair of fp32 -> one packed dword (2 x bf16) per piece
template <int NPL>
__device__ __forceinline__ void split_pair(float xe, float xo, unsigned (&pk)[NPL]) {
  constexpr unsigned SEL = 0x07060302u;  // D = {S1.b2, S1.b3, S0.b2, S0.b3}: upper halves of (even, odd)
  const unsigned ue = __float_as_uint(xe), uo = __float_as_uint(xo);
  pk[0] = __builtin_amdgcn_perm(uo, ue, SEL);
  const float re = xe - __uint_as_float(ue & 0xffff0000u), ro = xo - __uint_as_float(uo & 0xffff0000u);
  unsigned ure = __float_as_uint(re), uro = __float_as_uint(ro);
  if constexpr (NPL == 2) {  // last piece kept: round it to nearest instead of truncating
    ure += 0x8000u;
    uro += 0x8000u;
  }
  pk[1] = __builtin_amdgcn_perm(uro, ure, SEL);
  if constexpr (NPL == 3) {
    const float le = re - __uint_as_float(ure & 0xffff0000u), lo = ro - __uint_as_float(uro & 0xffff0000u);
    pk[2] = __builtin_amdgcn_perm(__float_as_uint(lo), __float_as_uint(le), SEL);
  }
}

// 8 consecutive-k fp32 values -> the three bf16x8 MFMA operand pieces
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, bf16x8 (&piece)[3]) {
  unsigned q0[3], q1[3], q2[3], q3[3];
  split_pair<3>(lo4[0], lo4[1], q0);
  split_pair<3>(lo4[2], lo4[3], q1);
  split_pair<3>(hi4[0], hi4[1], q2);
  split_pair<3>(hi4[2], hi4[3], q3);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) piece[pl] = __builtin_bit_cast(bf16x8, (u32x4){q0[pl], q1[pl], q2[pl], q3[pl]});
}

}  // namespace rnnt
