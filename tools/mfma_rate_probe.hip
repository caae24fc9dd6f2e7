// Probe: issue cycles per f32-input MFMA variant on one wave (4 independent accumulator chains), gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int n, unsigned long long* out, float* sink) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  {
    f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
  {
    f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    sink[64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
  {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    sink[128 + threadIdx.x] = c0[0] + c1[1];
  }
  {  // bf16 16x16x32 for reference
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 x, y; for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + threadIdx.x); y[i] = (short)0x3f80; }
    f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c3, 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    sink[192 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
}
int main() {
  unsigned long long* d; float* s; hipMalloc(&d, 64); hipMalloc(&s, 4096);
  const int n = 2000;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, n, d, s);
  unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  const char* names[4] = {"v_mfma_f32_4x4x1_16B_f32 (512 FLOP)", "v_mfma_f32_16x16x4_f32 (2048 FLOP)", "v_mfma_f32_32x32x2_f32 (4096 FLOP)", "v_mfma_f32_16x16x32_bf16 (16384 FLOP)"};
  const double fl[4] = {512, 2048, 4096, 16384};
  for (int i = 0; i < 4; ++i) printf("%-40s %.2f cycles/instr  -> %.1f FLOP/clk/SIMD\n", names[i], (double)h[i] / (4.0 * n), fl[i] / ((double)h[i] / (4.0 * n)));
  return 0;
}
