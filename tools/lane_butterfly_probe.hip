// Diagnostic: checks the lane butterflies of lstm_bwd5f_kernel (DPP rotations + gfx950 v_permlane16/32_swap) against a host loop.
//   hipcc --offload-arch=gfx950 -O3 tools/lane_butterfly_probe.hip -o tools/build/lane_butterfly_probe && tools/build/lane_butterfly_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false); }
// v_permlane16_swap / v_permlane32_swap (gfx950): rows 1, 3 of a <-> rows 0, 2 of b / lanes 32..63 of a <-> lanes 0..31 of b.  Inline
// asm: hipcc 7.2 models both results of __builtin_amdgcn_permlane*_swap as the first one (r[0] + r[1] became v_add v2, v2, v2).  The
// s_nop cover the VALU-write -> permlane-read wait states the hazard recogniser cannot see inside an asm block.
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xadd16(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  unsigned r[2] = {u, u};
  permlane16_swap(r[0], r[1]);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float xadd32(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  unsigned r[2] = {u, u};
  permlane32_swap(r[0], r[1]);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  float x = in[l];
  float a = x + __builtin_bit_cast(float, dpp_u32<0x124>(__builtin_bit_cast(unsigned, x)));
  out[l] = a;                                                                                  // x[l] + x[ror4]
  float b = a + __builtin_bit_cast(float, dpp_u32<0x128>(__builtin_bit_cast(unsigned, a)));
  out[64 + l] = b;                                                                             // sum over l, l+4, l+8, l+12 in the row
  float c = xadd16(b);
  out[128 + l] = c;                                                                            // + the row l ^ 16
  float d = xadd32(c);
  out[192 + l] = d;                                                                            // whole wave, same l & 3
  float q = x + __builtin_bit_cast(float, dpp_u32<0xB1>(__builtin_bit_cast(unsigned, x)));
  out[256 + l] = q;                                                                            // x[l] + x[l ^ 1]
  float r = x + __builtin_bit_cast(float, dpp_u32<0x4E>(__builtin_bit_cast(unsigned, x)));
  out[320 + l] = r;                                                                            // x[l] + x[l ^ 2]
}
int main() {
  float h[64], o[384], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 20)) + i;
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row = l & ~15, c = l & 15;
    float e0 = h[l] + h[row + ((c + 12) & 15)], e0b = h[l] + h[row + ((c + 4) & 15)];
    if (o[l] != e0 && o[l] != e0b) { printf("ror4 lane %d: %g (want %g or %g)\n", l, o[l], e0, e0b); ++bad; }
    float e1 = 0; for (int j = 0; j < 4; ++j) e1 += h[row + ((c + 4 * j) & 15)];
    if (o[64 + l] != e1) { printf("ror4+8 lane %d: %g want %g\n", l, o[64 + l], e1); ++bad; }
    float e2 = 0; for (int rr = 0; rr < 2; ++rr) for (int j = 0; j < 4; ++j) e2 += h[((row & 32) + 16 * rr) + ((c + 4 * j) & 15)];
    if (o[128 + l] != e2) { printf("+x16 lane %d: %g want %g\n", l, o[128 + l], e2); ++bad; }
    float e3 = 0; for (int j = 0; j < 64; ++j) if ((j & 3) == (l & 3)) e3 += h[j];
    if (o[192 + l] != e3) { printf("+x32 lane %d: %g want %g\n", l, o[192 + l], e3); ++bad; }
    if (o[256 + l] != h[l] + h[l ^ 1]) { printf("xor1 lane %d: %g want %g\n", l, o[256 + l], h[l] + h[l ^ 1]); ++bad; }
    if (o[320 + l] != h[l] + h[l ^ 2]) { printf("xor2 lane %d: %g want %g\n", l, o[320 + l], h[l] + h[l ^ 2]); ++bad; }
  }
  printf("lane butterflies: %d mismatches\n", bad);
  return bad != 0;
}
