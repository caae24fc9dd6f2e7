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
__device__ __forceinline__ float sigmoid_hw(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_hw(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace rnnt
