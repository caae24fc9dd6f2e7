// Issue cycles per instruction of the f16 / bf16 matrix-core instructions of gfx950 on ONE wave (4 independent accumulator chains,
// operands in registers, no memory traffic), then the whole-chip rate of each with 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_rate_probe.hip -o tools/mfma_f16_rate_probe && tools/mfma_f16_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512) k(int n, unsigned long long* out, float* sink) {
  f16x8 a, b;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f * i); x[i] = (__bf16)(0.001f * threadIdx.x + i); y[i] = (__bf16)(0.5f * i); }
  const unsigned long long t0 = clock64();
  float s = 0.f;
  if constexpr (KIND == 11) {   // one group of the LSTM backward: 24 MFMAs, 12 distinct A fragments x 2, 4 distinct B, 4 accumulators
    f16x8 A[12], Bv[4];
    for (int q = 0; q < 12; ++q) { A[q] = a; for (int i = 0; i < 8; ++i) A[q][i] += (_Float16)q; asm volatile("" : "+v"(A[q])); }
    for (int q = 0; q < 4; ++q) { Bv[q] = b; for (int i = 0; i < 8; ++i) Bv[q][i] += (_Float16)q; asm volatile("" : "+v"(Bv[q])); }
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = (f32x4){0, 0, 0, 0};
    unsigned long long burst = 0;
    for (int i = 0; i < n / 6; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long u0 = clock64();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks * 6 + j], Bv[2 * ks], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks * 6 + (j + 2) % 6], Bv[2 * ks + 1], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks * 6 + (j + 2) % 6], Bv[2 * ks], c[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      burst += clock64() - u0;
      __builtin_amdgcn_sched_barrier(0);
    }
    s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = burst;
  } else if constexpr (KIND == 9 || KIND == 10) {   // bursts: 24 back-to-back MFMAs (4 chains), then ~1.3 us (9) / nothing (10) of idling, repeated
    f16x8 a1 = a, a2 = a, a3 = a, b1 = b;
    for (int i = 0; i < 8; ++i) { a1[i] += (_Float16)1; a2[i] += (_Float16)2; a3[i] += (_Float16)3; b1[i] += (_Float16)1; }
    asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b1));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long burst = 0;
    for (int i = 0; i < n / 6; ++i) {
      if constexpr (KIND == 9) for (int z = 0; z < 48; ++z) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long u0 = clock64();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, r & 1 ? b1 : b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, r & 1 ? b1 : b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, r & 1 ? b1 : b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, r & 1 ? b1 : b, c3, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      burst += clock64() - u0;
      __builtin_amdgcn_sched_barrier(0);
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = burst;
  } else if constexpr (KIND == 7 || KIND == 8) {   // A operands (7) / A and B (8) pinned to AGPRs, accumulators in VGPRs
    f16x8 a1 = a, a2 = a, a3 = a, a0 = a, b0 = b;
    for (int i = 0; i < 8; ++i) { a1[i] += (_Float16)1; a2[i] += (_Float16)2; a3[i] += (_Float16)3; }
    asm volatile("" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3));
    if constexpr (KIND == 8) asm volatile("" : "+a"(b0));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b0, c3, 0, 0, 0);
      asm volatile("" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (KIND == 6) {   // accumulators pinned to AGPRs, 24 back-to-back MFMAs per iteration like one group of the LSTM backward
    f16x8 a1 = a, a2 = a, a3 = a, b1 = b;
    for (int i = 0; i < 8; ++i) { a1[i] += (_Float16)1; a2[i] += (_Float16)2; a3[i] += (_Float16)3; b1[i] += (_Float16)1; }
    asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b1));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b, c3, 0, 0, 0);
      asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (KIND == 4 || KIND == 5) {   // distinct A operands per chain (4) / distinct A and B (5): what a real kernel issues
    f16x8 a1 = a, a2 = a, a3 = a, b1 = b, b2 = b, b3 = b;
    for (int i = 0; i < 8; ++i) { a1[i] += (_Float16)1; a2[i] += (_Float16)2; a3[i] += (_Float16)3; b1[i] += (_Float16)1; b2[i] += (_Float16)2; b3[i] += (_Float16)3; }
    asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b1), "+v"(b2), "+v"(b3));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, KIND == 5 ? b1 : b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, KIND == 5 ? b2 : b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, KIND == 5 ? b3 : b, c3, 0, 0, 0);
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (KIND == 0 || KIND == 1) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
      if constexpr (KIND == 0) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c3, 0, 0, 0);
      }
    }
    s = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
    for (int i = 0; i < n; ++i) {
      if constexpr (KIND == 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
      }
    }
    s = c0[0] + c1[1];
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}

template <int KIND>
void run(const char* name, double flop_per_instr) {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 16);
  hipMalloc(&sink, 4);
  const int n = 20000;
  hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, n, out, sink);
  hipDeviceSynchronize();
  unsigned long long cyc = 0;
  hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
  printf("%-28s %6.2f cycles/instr (one wave)", name, (double)cyc / (4.0 * n));
  if (KIND == 9 || KIND == 10 || KIND == 11) {
    // the same kernel on the WHOLE chip, one wave per SIMD: cycles per MFMA inside the bursts of workgroup 0
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, n, out, sink);
    hipDeviceSynchronize();
    unsigned long long b2[2] = {0, 0};
    hipMemcpy(b2, out, 16, hipMemcpyDeviceToHost);
    printf("   in-burst, whole chip: %6.2f cycles/MFMA", (double)b2[1] / (4.0 * (n / 6) * 6));
  }
  for (int wps = 1; wps <= 2; ++wps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, n, out, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, n, out, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("   %d wave/SIMD: %7.1f TFLOP/s", wps, 256.0 * 4 * wps * 4.0 * n * flop_per_instr / (ms * 1e-3) / 1e12);
  }
  printf("\n");
}


int main() {
  run<0>("v_mfma_f32_16x16x32_f16", 16384.0);
  run<1>("v_mfma_f32_16x16x32_bf16", 16384.0);
  run<2>("v_mfma_f32_32x32x16_f16", 32768.0);
  run<3>("v_mfma_f32_32x32x16_bf16", 32768.0);
  run<4>("16x16x32_f16, 4 distinct A", 16384.0);
  run<5>("16x16x32_f16, distinct A, B", 16384.0);
  run<6>("16x16x32_f16, AGPR accumulators", 16384.0);
  run<7>("16x16x32_f16, A in AGPRs", 16384.0);
  run<8>("16x16x32_f16, A and B in AGPRs", 16384.0);
  run<9>("bursts of 24 + 1.3 us idle", 16384.0);
  run<10>("bursts of 24, no idle", 16384.0);
  run<11>("backward-like group of 24", 16384.0);
  return 0;
}
