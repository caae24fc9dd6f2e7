// Probe: sustained full-chip rate of v_mfma_f32_32x32x2_f32 on random data (all CUs, 1..3 waves per SIMD), with the
// in-kernel clock (shader cycles per 100 MHz real-time tick).  This is the practical fp32-MFMA roofline of the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(int n, float* sink, unsigned long long* clk) {
  float a = 0.5f + 0.001f * (threadIdx.x % 97), b = 1.0f - 0.002f * (threadIdx.x % 89);
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.1f * i; c1[i] = 0.2f; c2[i] = -0.1f * i; c3[i] = 0.3f; }
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    a = a * 0.999f + 0.0001f;  // keep data changing
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* s; unsigned long long* c; hipMalloc(&s, 4 * 256 * 4096); hipMalloc(&c, 16);
  for (int blocks_per_cu = 1; blocks_per_cu <= 3; ++blocks_per_cu) {
    const int n = 40000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 1000, s, c);  // warm
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, n, s, c);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 /*waves*/ * n * 4 /*mfma*/ * 4096.0;
    printf("%d wave(s)/SIMD: %.1f TFLOP/s  (%.2f ms), in-kernel clock %.0f MHz\n", blocks_per_cu, flops / ms / 1e9, ms, 100.0 * (double)h[0] / (double)h[1]);
  }
  return 0;
}
