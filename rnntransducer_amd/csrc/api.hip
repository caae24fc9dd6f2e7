// Error plumbing + version of librnnt_hip (C ABI in include/rnnt_hip.h).
#include "common.hpp"

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

extern "C" int rnnt_hip_version(void) { return RNNT_HIP_ABI_VERSION; }
extern "C" const char* rnnt_hip_last_error(void) { return rnnt::g_err; }
extern "C" int rnnt_hip_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}
