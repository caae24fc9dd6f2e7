// Probe: lane/register maps of v_mfma_f32_4x4x1_16B_f32 and v_mfma_f32_16x16x4_f32 on gfx950 (exact integer data).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  float a = 1.f, b = 1.f;
  if (mode == 0) a = (float)(l + 1);           // 4x4x1: which A lane feeds each output
  if (mode == 1) b = (float)(l + 1);           // 4x4x1: which B lane
  if (mode == 2) a = (float)(1 << (l >> 4)) * (float)((l & 15) + 1) ;   // 16x16x4: A contributions
  if (mode == 3) b = (float)(1 << (l >> 4)) * (float)((l & 15) + 1);
  f32x4 d;
  if (mode < 2) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  else d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); float h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
