// Error plumbing + version of librnnt_hip (C ABI in include/rnnt_hip.h).
#include "common.hpp"

#include <math.h>

#include <mutex>
#include <vector>

namespace rnnt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rnnt

namespace rnnt {
// ---- opt-in live profiler: HIP events recorded on the SAME stream the kernels are launched on ----
namespace {
struct ProfRec { int kind; double work; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_mu;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

ProfScope::ProfScope(int kind, double work, hipStream_t s) : kind_(kind), work_(work), stream_(s), start_(nullptr) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  start_ = take_event();
  if (start_) (void)hipEventRecord((hipEvent_t)start_, s);
}
ProfScope::~ProfScope() {
  if (!start_) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t b = take_event();
  if (b) (void)hipEventRecord(b, stream_);
  g_prof.push_back({kind_, work_, (hipEvent_t)start_, b});
}
}  // namespace rnnt

extern "C" int rnnt_hip_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(rnnt::g_prof_mu);
  rnnt::g_prof_on = on != 0;
  return RNNT_OK;
}

extern "C" int rnnt_hip_prof_collect(double* ms, double* work, int64_t* count, int nkinds) {
  using namespace rnnt;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < nkinds; ++k) { ms[k] = 0; work[k] = 0; count[k] = 0; }
  for (auto& r : g_prof) {
    if (r.a && r.b) {
      RNNT_CHECK_HIP(hipEventSynchronize(r.b));
      float t = 0.f;
      RNNT_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
      if (r.kind >= 0 && r.kind < nkinds) { ms[r.kind] += t; work[r.kind] += r.work; count[r.kind] += 1; }
    }
    if (r.a) g_pool.push_back(r.a);
    if (r.b) g_pool.push_back(r.b);
  }
  g_prof.clear();
  return RNNT_OK;
}

namespace rnnt {
namespace {
// torch.optim.AdamW step (decoupled weight decay) over flat fp32 buffers, same operation order as torch's single-tensor
// path: p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                          float wd, float step_size, float bc2_sqrt, float gscale,
                                                          const unsigned* __restrict__ guard) {
  // guard: the persistent recurrences' sticky status word (or nullptr).  Raised = a gradient of this step may be garbage:
  // skip the whole update (uniform branch), the host raises at its next status check.
  if (guard && *guard != 0u) return;
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i4), gg = *reinterpret_cast<const f32x4*>(g + i4) * gscale;
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i4), vv = *reinterpret_cast<f32x4*>(v + i4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= 1.0f - lr * wd;
      mm[e] += (gg[e] - mm[e]) * (1.0f - b1);
      vv[e] = vv[e] * b2 + (1.0f - b2) * gg[e] * gg[e];
      pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
    }
    *reinterpret_cast<f32x4*>(p + i4) = pp;
    *reinterpret_cast<f32x4*>(m + i4) = mm;
    *reinterpret_cast<f32x4*>(v + i4) = vv;
  } else {
    for (long i = i4; i < n; ++i) {
      float pp = p[i] * (1.0f - lr * wd);
      const float gg = g[i] * gscale;
      const float mm = m[i] + (gg - m[i]) * (1.0f - b1);
      const float vv = v[i] * b2 + (1.0f - b2) * gg * gg;
      pp -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
}
}  // namespace
}  // namespace rnnt

extern "C" int rnnt_hip_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int64_t step, void* stream) {
  return rnnt_hip_adamw_step_ex(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, 1.0f, nullptr, stream);
}

extern "C" int rnnt_hip_adamw_step_ex(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                      float eps, float weight_decay, int64_t step, float grad_scale, const uint32_t* guard,
                                      void* stream) {
  using namespace rnnt;
  RNNT_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adamw: bad arguments");
  RNNT_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adamw: buffers must be 16-byte aligned");
  if (n == 0) return RNNT_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  ProfScope prof(RNNT_K_MISC, 28.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)ceil_div(ceil_div(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long)n, lr, beta1, beta2, eps, weight_decay, (float)(lr / bc1), (float)sqrt(bc2), grad_scale, guard);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

extern "C" int rnnt_hip_version(void) { return RNNT_HIP_ABI_VERSION; }
extern "C" const char* rnnt_hip_last_error(void) { return rnnt::g_err; }
extern "C" int rnnt_hip_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}
