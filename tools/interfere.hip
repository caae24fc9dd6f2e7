// Stand-in loads for tools/interfere_probe.py: kernels that keep the XCDs NOT in `xcd_skip` busy with ONE kind of work while a
// persistent recurrence runs on the skipped XCDs — to tell apart what slows the recurrence down when a GEMM runs beside it:
//   mfma_spin : matrix-core work only (no memory traffic beyond the final store)  -> clocks / power
//   mem_stream: HBM streaming only (reads `bytes` per pass, no matrix-core work)  -> memory system
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/interfere.hip -o tools/libinterfere.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool skipped(unsigned xcd_skip) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return (xcd_skip >> (xcc & 7u)) & 1u;
}

__global__ void __launch_bounds__(512) mfma_spin_kernel(unsigned xcd_skip, int iters, float* out) {
  if (skipped(xcd_skip)) return;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(256) mem_stream_kernel(unsigned xcd_skip, const f32x4* __restrict__ src, long n16, int passes, float* out) {
  if (skipped(xcd_skip)) return;
  float s = 0.f;
  for (int p = 0; p < passes; ++p)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
      const f32x4 v = __builtin_nontemporal_load(src + i);
      s += v[0] + v[1] + v[2] + v[3];
    }
  if (s == 123.456f) out[0] = s;
}

extern "C" int interfere_mfma_spin(unsigned xcd_skip, int iters, float* out, void* stream) {
  hipLaunchKernelGGL(mfma_spin_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, xcd_skip, iters, out);
  return (int)hipGetLastError();
}
extern "C" int interfere_mem_stream(unsigned xcd_skip, const void* src, long bytes, int passes, float* out, void* stream) {
  hipLaunchKernelGGL(mem_stream_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, xcd_skip, (const f32x4*)src, bytes / 16, passes, out);
  return (int)hipGetLastError();
}
