// fp32 GEMM on the f16 matrix cores of gfx950 through "half-pair" (hp) operands: three v_mfma_f32_16x16x32_f16 per fp32
// product block instead of six bf16 ones (gemm.hip), operands delivered to LDS by LDS-DMA with no VALU in the main loop.
//
// Replaces the BLAS calls behind the BIG products of the hot path: the hoisted LSTM input projection W_ih.x_t over all
// frames (networks/encoder.py:67-75,99) and its three backward products dX = dG.W_ih, dW_ih = dG^T.X, dW_hh = dG^T.h_prev
// (autograd of the same lines) — 95 % of the step's GEMM FLOPs at BASELINE configs[1].  Small / oddly shaped products
// (out_proj, joint pre-GEMMs, K < 256) stay on gemm.hip.
//
// hp format of an fp32 matrix x (rows x K), scaled PER ROW: amax_r = max_k |x[r][k]|, s_r = 2^(14 - floor(log2 amax_r)),
//   v = x * s_r (|v| < 2^15),  hi = fp16_rn(v),  lo = fp16_rn(v - hi)  =>  v = hi + lo + e,  |e| <= 2^-23 |v|  (11 + 1 + 11 bits)
//   stored row-major, K padded to 32, per (row, 32-k block) ONE 128-byte line: 32 x hi | 32 x lo; amax_r (fp32 bits) in a
//   side array.  Row scales factor out of a dot product (C[m][n] = s_m^-1 s_n^-1 sum_k a'b'), so every row keeps full relative
//   precision whatever the other rows hold; inside a row, values more than 2^17 below the row's amax lose relative (not absolute)
//   precision: |e| <= 2^-40 amax_r — below the fp32 rounding of any sum the row's large elements take part in.  a.b ~= s_a^-1 s_b^-1 (a_lo b_hi + a_hi b_lo + a_hi b_hi): the dropped a_lo b_lo is
// <= 2^-22 |a b|, typically 2^-25; products and sums are exact / fp32-accumulated in the MFMA.  Measured against fp64:
// tests/test_gpu_gemm.py::test_gemm_hp_*.
//
// Kernel (NT form only: both operands k-contiguous; transposed operands are produced as such by the split kernels):
//   256 x 256 x 32 tile, 512 threads = 8 waves as 2 (M) x 4 (N), 128 x 64 of C per wave = 8 x 4 blocks of 16 x 16,
//   96 MFMAs per wave per K-tile; 2 x 64 KB LDS stages filled by buffer_load_dwordx4 ... lds (8 per thread per K-tile),
//   LDS rows of 128 B (hi | lo) with the 16-byte slot s of row r stored at slot s ^ ((r >> 1) & 7): every ds_read_b128 of a
//   16x16x32 operand fragment is bank-conflict-free; the swizzle is applied on the per-lane SOURCE address (LDS-DMA writes
//   lane-linear).
#include "common.hpp"

#include <stdlib.h>

#include <type_traits>

namespace rnnt {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HP_BM = 256, HP_BN = 256, HP_BK = 32;
constexpr int HP_STAGE = (HP_BM + HP_BN) * 128;  // bytes per K-tile stage (A rows | B rows, 128 B each)
constexpr int HP_RSRC = 0x00027000;

__device__ __forceinline__ float hp_scale_from_amax(unsigned amax_bits) {
  // 2^(14 - e) with e = floor(log2 amax); amax == 0 (or denormal) -> 1
  int eb = (int)((amax_bits >> 23) & 255u);
  if (eb == 0) return 1.0f;
  eb = eb < 15 ? 15 : eb;  // |x| < 2^-112 everywhere: keep both factors finite (scale * inv_scale == 1 exactly either way)
  return __uint_as_float((unsigned)(268 - eb) << 23);
}
__device__ __forceinline__ float hp_inv_scale_from_amax(unsigned amax_bits) {
  int eb = (int)((amax_bits >> 23) & 255u);
  if (eb == 0) return 1.0f;
  eb = eb < 15 ? 15 : eb;
  return __uint_as_float((unsigned)(eb - 14) << 23);  // 2^(e - 14), e = eb - 127
}

// ------------------------------------------------------------------------------------------------------------------
// column maxima of |x| over a (rows x C) view with row stride ld, as bit patterns of non-negative floats (atomicMax on unsigned
// orders them like the floats; max is order-independent, so the result is deterministic).  out must be zeroed.
// grid (ceil(C/256), row chunks)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hp_colmax_kernel(const float* __restrict__ x, long rows, int C, long ld, long rows_per_chunk,
                                                        unsigned* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  unsigned m = 0;
  for (long r = r0; r < r1; ++r) m = max(m, __float_as_uint(x[r * ld + c]) & 0x7fffffffu);
  if (m != 0) atomicMax(out + c, m);
}

__device__ __forceinline__ void hp_split8(const float (&x)[8], float scale, u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e] * scale;
    const _Float16 a = (_Float16)v;   // round to nearest even
    h[e] = a;
    l[e] = (_Float16)(v - (float)a);  // exact residual, rounded once
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}

// x (rows x K, row stride ld) -> planes[rows][Kp/32][hi 32 | lo 32] + amax[rows]; ONE WAVE per row: pass 1 row maximum
// (wave reduce), pass 2 scale + split (the row is re-read from L1/L2)
// rowidx (optional): only the `rows` listed rows are converted, each IN PLACE (source row rowidx[i] -> plane row rowidx[i], amax[rowidx[i]]):
// the valid frames of a ragged batch; the GEMM then gathers exactly those plane rows (HpGemmK::a_rowidx).
__global__ void __launch_bounds__(256) hp_split_kernel(const float* __restrict__ x, long rows, int K, long ld, unsigned* __restrict__ amax,
                                                       char* __restrict__ out, const int* __restrict__ rowidx) {
  const int Kp = (K + 31) & ~31, cpr = Kp >> 3;
  const int lane = threadIdx.x & 63;
  const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  for (long ri = (long)blockIdx.x * 4 + (threadIdx.x >> 6); ri < rows; ri += (long)gridDim.x * 4) {
    const long r = rowidx ? (long)rowidx[ri] : ri;
    const float* row = x + r * ld;
    unsigned m = 0;
    for (int c = lane; c < cpr; c += 64) {
      const int k = 8 * c;
      if (vec && k + 7 < K) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(row + k), b = *reinterpret_cast<const u32x4*>(row + k + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m = max(m, max(a[e] & 0x7fffffffu, b[e] & 0x7fffffffu));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k + e < K) m = max(m, __float_as_uint(row[k + e]) & 0x7fffffffu);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) amax[r] = m;
    const float scale = hp_scale_from_amax(m);
    for (int c = lane; c < cpr; c += 64) {
      const int k = 8 * c;
      float v[8];
      if (vec && k + 7 < K) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + k), b = *reinterpret_cast<const f32x4*>(row + k + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? row[k + e] : 0.f;
      }
      u32x4 hi, lo;
      hp_split8(v, scale, hi, lo);
      char* dst = out + (r * (Kp >> 5) + (c >> 2)) * 128 + 16 * (c & 3);
      *reinterpret_cast<u32x4*>(dst) = hi;
      *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
  }
}

// transposed: x is (Ksrc rows x >= R cols, row stride ld); planes row r (= source column c0 + r), contraction index k in [0, K):
// value x[k + shift][c0 + r] (0 when k + shift is outside [0, Ksrc)), scaled by amax[r] (the source column's maximum: hp_colmax).
// Workgroup = 32 k x 256 source columns through LDS.
// kidx (optional): contraction index k takes source row kidx[k] + shift instead of k + shift (the valid frames of a ragged batch,
// packed along the contraction: K = their number).
__global__ void __launch_bounds__(256) hp_split_t_kernel(const float* __restrict__ x, int R, int K, long ld, int Ksrc, int shift,
                                                         const unsigned* __restrict__ amax, char* __restrict__ out,
                                                         const int* __restrict__ kidx) {
  __shared__ float tile[32][257];
  const int kb = blockIdx.x, r0 = blockIdx.y * 256;
  const int tid = threadIdx.x;
  // load: pass q covers source rows 4q..4q+3, each row 64 threads x 4 columns
  const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int kk = 4 * q + (tid >> 6);
    const int kdst = kb * 32 + kk;
    const int ksrc = (kidx ? (kdst < K ? kidx[kdst] : 0) : kdst) + shift;
    const int c = 4 * (tid & 63);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kdst < K && ksrc >= 0 && ksrc < Ksrc) {
      const float* src = x + (long)ksrc * ld + r0 + c;
      if (vec && r0 + c + 3 < R) {
        v = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r0 + c + e < R) v[e] = src[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[kk][c + e] = v[e];
  }
  __syncthreads();
  // store: thread -> (output row r0 + tid/4 + 64*pass, chunk tid%4)
  const int Kp32 = (K + 31) >> 5;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rl = (tid >> 2) + 64 * ps, ch = tid & 3;
    if (r0 + rl < R) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[8 * ch + e][rl];
      u32x4 hi, lo;
      hp_split8(v, hp_scale_from_amax(amax[r0 + rl]), hi, lo);   // per output row (= source column) scale
      char* dst = out + ((long)(r0 + rl) * Kp32 + kb) * 128 + 16 * ch;
      *reinterpret_cast<u32x4*>(dst) = hi;
      *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
  }
}

// Both orientations of one matrix in ONE pass over it (dG of an LSTM layer: row-major planes for dX = dG . W_ih, transposed planes for
// dW = dG^T . X): x (M x C, row stride ld) -> out_rm[M][Cp/32] lines scaled by rowmax[r]  and  out_t[C][Mp/32] lines scaled by
// colmax[c]; both tables are GIVEN (the backward recurrence leaves them).  Workgroup = 32 rows x 256 columns: every thread loads 8
// consecutive values of a row, emits their row-major half-line from registers and parks them in LDS for the transposed store.
// rowidx (optional, M = its length): source row i of the walk is rowidx[i]; its row-major half-lines go to plane row rowidx[i] (in place,
// scaled by rowmax[rowidx[i]]), the transposed planes are packed along the contraction (index i, M of them).
__global__ void __launch_bounds__(256) hp_split_both_kernel(const float* __restrict__ x, int M, int C, long ld, const unsigned* __restrict__ rowmax,
                                                            const unsigned* __restrict__ colmax, char* __restrict__ out_rm, char* __restrict__ out_t,
                                                            const int* __restrict__ rowidx) {
  __shared__ float tile[32][257];
  const int kb = blockIdx.x, c0 = blockIdx.y * 256;
  const int tid = threadIdx.x;
  const int Cp = (C + 31) & ~31;
  const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * q + (tid >> 5), col = 8 * (tid & 31);
    const int ri = kb * 32 + row, c = c0 + col;
    const int r = (rowidx && ri < M) ? rowidx[ri] : ri;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ri < M && c < C) {
      const float* src = x + (long)r * ld + c;
      if (vec && c + 7 < C) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c + e < C) v[e] = src[e];
      }
    }
    if (ri < M && c < Cp) {
      u32x4 hi, lo;
      hp_split8(v, hp_scale_from_amax(rowmax[r]), hi, lo);
      char* dst = out_rm + ((long)r * (Cp >> 5) + (c >> 5)) * 128 + 16 * ((c >> 3) & 3);
      *reinterpret_cast<u32x4*>(dst) = hi;
      *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[row][col + e] = v[e];
  }
  __syncthreads();
  const int Mp32 = (M + 31) >> 5;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rl = (tid >> 2) + 64 * ps, ch = tid & 3;
    if (c0 + rl < C) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[8 * ch + e][rl];
      u32x4 hi, lo;
      hp_split8(v, hp_scale_from_amax(colmax[c0 + rl]), hi, lo);
      char* dst = out_t + ((long)(c0 + rl) * Mp32 + kb) * 128 + 16 * ch;
      *reinterpret_cast<u32x4*>(dst) = hi;
      *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
struct HpGemmK {
  int M, N, nkt;             // nkt = K-tiles (32 k each) of the whole contraction
  const char* A; unsigned a_pitch, a_bytes;   // hp planes, bytes per row = nkt * 128
  const char* B; unsigned b_pitch, b_bytes;
  const unsigned* a_amax; const unsigned* b_amax;
  float* C; int c_div; long c_so, c_si;
  const float* bias;
  unsigned flags;            // RNNT_GEMM_ACCUM
  int splits, kt_per_split;  // blockIdx.y = split z handles K-tiles [z * kt_per_split, ...)
  float* slab;               // splits > 1: slab[z][M][N]
  int tiles_m, tiles_n;
  int group_m;               // > 0: tiles walked in bands of group_m tile rows (see hp_grouped_tile); 0: plain column-major order
  const int* a_rowidx;       // optional: row m of the product reads plane row (and amax entry) a_rowidx[m] of A — the valid frames of a
  const int* c_rowidx;       // ragged batch gathered by the operand fetch; and writes output row c_rowidx[m] (scatter).  nullptr: m
};

// same bijective XCD remap idea as gemm.hip: consecutive tiles of one XCD share operand panels through its L2
__device__ __forceinline__ int hp_xcd_remap(int bid, int n) {
  const int q = n / 8, r = n % 8, xcd = bid % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// one 256 x 256 output tile (`bid` = column-major tile index) over K-tiles [z * kt_per_split, ...) of problem p
__device__ __forceinline__ void hp_tile256(const HpGemmK& p, const int bid, const int z, char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int m0 = (bid % p.tiles_m) * HP_BM, n0 = (bid / p.tiles_m) * HP_BN;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, (int)p.a_bytes, HP_RSRC);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, (int)p.b_bytes, HP_RSRC);

  // LDS-DMA piece j of this wave: tile rows (8j + wave) * 8 .. + 7, lane -> (row lane>>3, LDS slot lane&7), which holds source
  // slot (lane&7) ^ ((row >> 1) & 7) = (lane&7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2))
  unsigned va[4], vb[4];
  {
    const int src_slot = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (8 * j + wave) * 8 + (lane >> 3);
      int ma = min(m0 + row, p.M - 1);
      const int nb = min(n0 + row, p.N - 1);  // rows past the edge re-read the last row (never stored)
      if (p.a_rowidx) ma = p.a_rowidx[ma];
      va[j] = (unsigned)ma * p.a_pitch + 16u * src_slot;
      vb[j] = (unsigned)nb * p.b_pitch + 16u * src_slot;
    }
  }
  auto stage = [&](int buf, int kt) {
    char* base = lds + buf * HP_STAGE;
    const int koff = kt * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (8 * j + wave) * 1024), 16, va[j], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + HP_BM * 128 + (8 * j + wave) * 1024), 16, vb[j], koff, 0, 0);
    }
  };

  // operand fragment of a 16-row block: lane -> row lane&15, k 8*(lane>>4)..+7 of plane pl: slot (4 pl + (lane>>4)) ^ ((row>>1)&7)
  const int frag0 = (lane & 15) * 128 + 16 * ((lane >> 4) ^ ((lane & 15) >> 1));   // plane hi; plane lo = frag0 ^ 64
  const int a_base = wr * (128 * 128) + frag0;
  const int b_base = HP_BM * 128 + wc * (64 * 128) + frag0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kt0 = z * p.kt_per_split;
  const int nk = min(p.nkt - kt0, p.kt_per_split);

#ifdef HP_PIPE
  // Software-pipelined main loop — build variant only (-DHP_PIPE=1), NOT the default: it measured within 1 % of the plain loop below
  // (profiles/r03_gemm_hp_bound_probe.txt: dX 610 vs 617 us, proj 755 vs 741, dW 589 vs 592).  The probe variants of the same file
  // say why: the MFMAs alone run at 0.83 of the nominal f16 peak (power-limited clock); with the operand fragments read from LDS at
  // 0.61-0.63 WHETHER OR NOT the reads are issued a quarter tile ahead (so it is not exposed LDS latency: every ds_read_b128 costs the
  // SIMD about 20 cycles of MFMA issue while its 1 KB lands in the register file), and the operand fetch alone (LDS-DMA, no MFMA) takes
  // 0.72 of the full kernel's time at 77 % L2 hits (one 64 KB round trip per K-tile and CU: a 2-stage ring cannot keep more in flight).
  // Every ds_read_b128 is issued at least 12 MFMAs (usually a whole quarter = 24) ahead of its first use:
  //   quarter  operands (A half, B half)   loads issued during it
  //   Q0       a(mq 0), set0 = b(nq 0)     set1 <- b(nq 1)
  //   Q1       a(mq 0), set1               a[i] <- A(mq 1) block i, as soon as block i's MFMAs are issued
  //   Q2       a(mq 1), set1               set0 <- b(nq 0) again;  then vmcnt(0) + lgkmcnt(0) + BARRIER (the only one per K-tile)
  //   Q3       a(mq 1), set0               LDS-DMA of K-tile kt + 2 into THIS tile's (now fully read) stage;  set1 <- next tile's b(nq 0),
  //                                        a[i] <- next tile's A(mq 0) block i — from the other stage, whose DMA the barrier waited for
  // The two B register sets swap roles every K-tile (the loop body is instantiated for both parities).  Every accumulator still sees
  // its K-tiles and, inside one, its three products in the same order: results are bitwise those of the plain loop (HP_NO_PIPE).
  f16x8 a[4][2], bs[2][2][2];
  auto ld_a = [&](const char* sb, int mq, int i) {
    a[i][0] = *reinterpret_cast<const f16x8*>(sb + (a_base + (4 * mq + i) * 2048));
    a[i][1] = *reinterpret_cast<const f16x8*>(sb + ((a_base ^ 64) + (4 * mq + i) * 2048));
  };
  auto ld_b = [&](int set, const char* sb, int nq) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bs[set][j][0] = *reinterpret_cast<const f16x8*>(sb + (b_base + (2 * nq + j) * 2048));
      bs[set][j][1] = *reinterpret_cast<const f16x8*>(sb + ((b_base ^ 64) + (2 * nq + j) * 2048));
    }
  };
  // blocks i0, i0 + 1 of A half mq against B half nq (register set `set`): 12 MFMAs, four independent accumulators between dependent ones
  auto mma2 = [&](int mq, int nq, int set, int i0) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 c = acc[4 * mq + i][2 * nq + j];
          if (t == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], bs[set][j][0], c, 0, 0, 0);   // smallest terms first
          else if (t == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], bs[set][j][1], c, 0, 0, 0);
          else c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], bs[set][j][0], c, 0, 0, 0);
          acc[4 * mq + i][2 * nq + j] = c;
        }
  };
#define HP_PIN() __builtin_amdgcn_sched_barrier(0)   /* nothing moves across: the loads stay where the table above puts them */

  stage(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nk > 1) stage(1, kt0 + 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) ld_a(lds, 0, i);
  ld_b(0, lds, 0);
  HP_PIN();

  auto ktile = [&](auto parity, const int kt) {
    constexpr int S0 = decltype(parity)::value, S1 = S0 ^ 1;   // B register sets: S0 holds nq 0 on entry
    const char* sb = lds + S0 * HP_STAGE;                       // stage of this K-tile: kt0-relative parity == S0 by construction
    const char* nb = lds + S1 * HP_STAGE;
    // Q0  (the loads of set1 sit BEHIND the first 12 MFMAs: at the loop head hipcc waits with lgkmcnt(0) — it cannot count what the
    //      previous iteration left in flight — and loads issued in front of that wait would be waited for at once)
    mma2(0, 0, S0, 0);
    HP_PIN();            // (blocks 2, 3 of A arrived 12 MFMAs later than blocks 0, 1: their MFMAs stay behind those of 0, 1)
    ld_b(S1, sb, 1);
    HP_PIN();
    mma2(0, 0, S0, 2);
    HP_PIN();
    // Q1
    mma2(0, 1, S1, 0);
    HP_PIN();
    ld_a(sb, 1, 0); ld_a(sb, 1, 1);
    HP_PIN();
    mma2(0, 1, S1, 2);
    HP_PIN();
    ld_a(sb, 1, 2); ld_a(sb, 1, 3);
    ld_b(S0, sb, 0);
    HP_PIN();
    // Q2
    mma2(1, 1, S1, 0);
    HP_PIN();
    mma2(1, 1, S1, 2);
    HP_PIN();
    __syncthreads();   // vmcnt(0): the LDS-DMA of K-tile kt + 1 has landed; lgkmcnt(0): every read of this stage is done
    HP_PIN();
#ifndef HP_DBG_NO_DMA
    if (kt + 2 < nk) stage(S0, kt0 + kt + 2);
#endif
    // Q3 (+ the first fragments of K-tile kt + 1; behind the last K-tile these read stale LDS into dead registers)
    ld_b(S1, nb, 0);
    HP_PIN();
    mma2(1, 0, S0, 0);
    HP_PIN();
    ld_a(nb, 0, 0); ld_a(nb, 0, 1);
    HP_PIN();
    mma2(1, 0, S0, 2);
    HP_PIN();
    ld_a(nb, 0, 2); ld_a(nb, 0, 3);
    HP_PIN();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) ktile(std::integral_constant<int, 1>{}, kt + 1);
  }
#undef HP_PIN
#else
  stage(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // (measured in one process, tools/gemm_hp_ab.py: issuing the LDS-DMA of waves 4-7 half a tile later than that of their SIMD
    // partners 0-3 is 5 % SLOWER, s_setprio around the MFMA clusters 1 % slower, 256 x 128 tiles with a 3-stage ring 13 % slower)
#ifndef HP_DBG_NO_DMA   // (diagnostic builds: tools/gemm_hp_bound_probe.sh times the loop without its operand fetch / without its MFMAs)
    if (kt + 1 < nk) stage(cur ^ 1, kt0 + kt + 1);   // lands under this tile's 96 MFMAs
#endif
    const char* sb = lds + cur * HP_STAGE;
    f16x8 a[4][2], b[2][2];
#ifdef HP_DBG_NO_LDSREAD   // (diagnostic: operand fragments stay whatever the registers hold — the MFMA pipe alone)
    auto load_a = [&](int mq) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(a[i][0]), "+v"(a[i][1])); }
    };
    auto load_b = [&](int nq) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(b[j][0]), "+v"(b[j][1])); }
    };
#else
    auto load_a = [&](int mq) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i][0] = *reinterpret_cast<const f16x8*>(sb + (a_base + (4 * mq + i) * 2048));
        a[i][1] = *reinterpret_cast<const f16x8*>(sb + ((a_base ^ 64) + (4 * mq + i) * 2048));
      }
    };
    auto load_b = [&](int nq) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b[j][0] = *reinterpret_cast<const f16x8*>(sb + (b_base + (2 * nq + j) * 2048));
        b[j][1] = *reinterpret_cast<const f16x8*>(sb + ((b_base ^ 64) + (2 * nq + j) * 2048));
      }
    };
#endif
    auto mma = [&](int mq, int nq) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 c = acc[4 * mq + i][2 * nq + j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[4 * mq + i][2 * nq + j] = c;
        }
    };
#ifndef HP_DBG_NO_MFMA
    load_a(0); load_b(0); mma(0, 0);
    load_b(1); mma(0, 1);
    load_a(1); mma(1, 1);
    load_b(0); mma(1, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);  // keep all 96 MFMAs in front of the wait (hipcc otherwise sinks half of them behind the barrier)
#ifndef HP_DBG_NO_BARRIER
    __syncthreads();   // drains the LDS-DMA of the next stage (vmcnt(0)) and fences this stage's reads
#endif
  }

#endif
  // epilogue: D block (i, j): lane -> rows 4*(lane>>4) + reg, column lane&15
  const int mode = p.splits > 1 ? 0 : ((p.flags & RNNT_GEMM_ACCUM) ? 2 : 1);  // uniform: slab | store | accumulate
  float* slab = p.splits > 1 ? p.slab + (long)z * p.M * p.N : nullptr;
  float bias_v[4], sb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + (lane & 15);
    bias_v[j] = (mode != 0 && p.bias && n < p.N) ? p.bias[n] : 0.f;
    sb[j] = hp_inv_scale_from_amax(p.b_amax[min(n, p.N - 1)]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = m0 + wr * 128 + i * 16 + 4 * (lane >> 4) + reg;
      const bool mok = m < p.M;
      const int mc = mok ? m : 0;
      const float sa = hp_inv_scale_from_amax(p.a_amax[p.a_rowidx ? p.a_rowidx[mc] : mc]);
      const int mo = p.c_rowidx ? p.c_rowidx[mc] : mc;
      float* crow = mode == 0 ? slab + (long)mc * p.N : p.C + (long)(mo / p.c_div) * p.c_so + (long)(mo % p.c_div) * p.c_si;
      float old[4] = {0.f, 0.f, 0.f, 0.f};
      if (mode == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wc * 64 + j * 16 + (lane & 15);
          if (mok && n < p.N) old[j] = crow[n];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane & 15);
        if (mok && n < p.N) crow[n] = acc[i][j][reg] * sa * sb[j] + bias_v[j] + old[j];
      }
    }
  }
}

// Walk order of an XCD's contiguous share of the tiles.  Column-major order puts 32 tiles of ONE tile column in flight on an XCD:
// 32 A panels + 1 B panel pass through its L2 for 32 tiles, and A is fetched tiles_n times over the launch (PMC: 1.4 GB read per
// c2 launch against 0.15-0.55 GB of operands).  In bands of group_m tile rows (m fastest inside a band, then n) the 32 tiles in
// flight are group_m x 32/group_m: 8 + 4 panels instead of 33.  Bijective on [0, tiles_m * tiles_n); returns the column-major index.
__device__ __forceinline__ int hp_grouped_tile(int idx, int tiles_m, int tiles_n, int group_m) {
  const int band = group_m * tiles_n;
  const int g = idx / band, r = idx - g * band;
  const int m_first = g * group_m;
  const int gm = min(tiles_m - m_first, group_m);
  const int n = r / gm, m = m_first + (r - n * gm);
  return n * tiles_m + m;
}

__global__ void __launch_bounds__(512, 1) gemm_hp_kernel(const HpGemmK p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int bid = hp_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  if (p.group_m > 0) bid = hp_grouped_tile(bid, p.tiles_m, p.tiles_n, p.group_m);
  hp_tile256(p, bid, blockIdx.y, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// Grouped, queue-driven form for work that runs BESIDE a persistent recurrence (the weight-gradient products of layer l under
// the reverse-time recurrence of layer l-1, lstm.hip): up to 4 problems in one launch, 256 resident workgroups that draw
// (problem, tile, K-split) units from one atomic counter until it runs dry.  `xcd_skip`: bit x set = workgroups that find
// themselves on XCD x leave at once (the recurrence owns those XCDs: its workgroups exchange through that XCD's L2 and must not
// share SIMDs or L2 with a GEMM); the remaining XCDs do all units.  Which workgroup computes which unit varies from run to run,
// the arithmetic of a unit does not: results are bitwise reproducible.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HPQ_MAX = 4;
struct HpGemmQ {
  HpGemmK prob[HPQ_MAX];
  int unit_end[HPQ_MAX];   // prefix sums of tiles * splits
  int nprob;
  unsigned xcd_skip;
  unsigned* counter;       // 16 words, zeroed before the launch: [0..7] one queue per XCD, [8] units completed, [9] "incomplete" (hpq_check_kernel)
};

// Unit u of a problem: n-tile fastest, then K-split, then m-tile — neighbours in the queue share the A panel (same rows, same
// K range) and, every tiles_n units, the B panels.  Every participating XCD owns one contiguous share of the units (its 32
// workgroups walk it front to back: what they have in flight at any time re-uses a handful of panels through that XCD's L2;
// drawing from ONE queue across XCDs measured 2.95 ms against a 2.0 ms estimate for a c2 layer: every tile fetched both panels
// from HBM); an XCD that runs dry takes units from the others' shares.
__global__ void __launch_bounds__(512, 1) gemm_hpq_kernel(const HpGemmQ q) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  if ((q.xcd_skip >> xcc) & 1u) return;
  const unsigned allow = ~q.xcd_skip & 0xffu;
  const int nq = __builtin_popcount(allow);
  const int mine = __builtin_popcount(allow & ((1u << xcc) - 1u));
  int* next = reinterpret_cast<int*>(lds + 2 * HP_STAGE);
  const int total = q.unit_end[q.nprob - 1];
  for (int hop = 0; hop < nq; ++hop) {
    const int qi = (mine + hop) % nq;
    const int lo = (int)((long)total * qi / nq), hi = (int)((long)total * (qi + 1) / nq);
    while (true) {
      if (threadIdx.x == 0) {
        const int got = lo + (int)atomicAdd(q.counter + qi, 1u);
        *next = got;
        // units drawn (= units done once the kernel has ended), read by hpq_check_kernel behind this launch.  Counted HERE, inside the
        // one divergent region of the loop: a second `if (threadIdx.x == 0)` region at the loop's end made hipcc 7.2 restructure the loop
        // so that lanes != 0 re-entered the barriers without lane 0 ever refreshing *next (an endless loop on the same unit).
        if (got < hi) atomicAdd(q.counter + 8, 1u);
      }
      __syncthreads();
      const int u = __builtin_amdgcn_readfirstlane(*next);
      __syncthreads();
      if (u >= hi) break;
      int pi = 0;
      while (u >= q.unit_end[pi]) ++pi;
      const HpGemmK& p = q.prob[pi];
      const int local = u - (pi ? q.unit_end[pi - 1] : 0);
      const int tn = local % p.tiles_n, z = (local / p.tiles_n) % p.splits, tm = local / (p.tiles_n * p.splits);
      hp_tile256(p, tn * p.tiles_m + tm, z, lds);
    }
  }
}

// Self-check of the queue-driven launch: every unit must have been drawn by SOME workgroup.  With `xcd_skip` set that rests on the
// device exposing the XCDs the mask leaves (a partitioned device, a CU-masked queue or another XCC numbering could make every
// workgroup leave): then nothing was written, and instead of un-permuting stale slabs into the gradients with rc = 0 this raises
// counter[9] and the caller's sticky status word (2), which the guarded AdamW update and FlatAdamW.step() act on.
__global__ void hpq_check_kernel(unsigned* counter, unsigned total, unsigned* status) {
  if (counter[8] != total) {
    counter[9] = 1u;
    if (status) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 128 x 32 tile, 8 waves as 4 (M) x 2 (N), 64 x 64 of C per wave (4 x 4 blocks, 48 MFMAs per K-tile), THREE 48 KB LDS stages:
// the LDS-DMA of K-tile t+2 is issued before tile t is multiplied and is only waited for (counted vmcnt, raw s_barrier: a
// __syncthreads() would drain it) at the end of tile t+1 — two tiles of MFMA time to land instead of one.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HP3_BM = 256, HP3_BN = 128, HP3_STAGE = (HP3_BM + HP3_BN) * 128, HP3_NST = 3;

__global__ void __launch_bounds__(512, 1) gemm_hp3_kernel(const HpGemmK p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int bid = hp_xcd_remap(blockIdx.x, ntiles);
  const int m0 = (bid % p.tiles_m) * HP3_BM, n0 = (bid / p.tiles_m) * HP3_BN;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, (int)p.a_bytes, HP_RSRC);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, (int)p.b_bytes, HP_RSRC);

  // LDS-DMA pieces: A rows (8j + wave) * 8 .. + 7 for j = 0..3, B rows (8j + wave) * 8 .. + 7 for j = 0..1
  unsigned va[4], vb[2];
  {
    const int src_slot = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (8 * j + wave) * 8 + (lane >> 3);
      va[j] = (unsigned)min(m0 + row, p.M - 1) * p.a_pitch + 16u * src_slot;
      if (j < 2) vb[j] = (unsigned)min(n0 + row, p.N - 1) * p.b_pitch + 16u * src_slot;
    }
  }
  const int kt0 = blockIdx.y * p.kt_per_split;
  const int nk = min(p.nkt - kt0, p.kt_per_split);
  auto stage = [&](int buf, int kt) {   // kt >= nk: offsets beyond the planes read zeros (keeps the vmcnt arithmetic uniform)
    char* base = lds + buf * HP3_STAGE;
    const int koff = kt < nk ? (kt0 + kt) * 128 : 0x7ff00000;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (8 * j + wave) * 1024), 16, va[j], koff, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + HP3_BM * 128 + (8 * j + wave) * 1024), 16, vb[j], koff, 0, 0);
  };

  const int frag0 = (lane & 15) * 128 + 16 * ((lane >> 4) ^ ((lane & 15) >> 1));
  const int a_base = wr * (64 * 128) + frag0;
  const int b_base = HP3_BM * 128 + wc * (64 * 128) + frag0;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  stage(1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage 0 landed (this wave's share); the barrier makes it everyone's
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int nxt2 = cur == 0 ? 2 : cur - 1;          // (cur + 2) % 3: the buffer read during the previous iteration
    stage(nxt2, kt + 2);
    const char* sb = lds + cur * HP3_STAGE;
    f16x8 a[4][2], b[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i][0] = *reinterpret_cast<const f16x8*>(sb + (a_base + i * 2048));
      a[i][1] = *reinterpret_cast<const f16x8*>(sb + ((a_base ^ 64) + i * 2048));
      b[i][0] = *reinterpret_cast<const f16x8*>(sb + (b_base + i * 2048));
      b[i][1] = *reinterpret_cast<const f16x8*>(sb + ((b_base ^ 64) + i * 2048));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // tile kt+1 landed; tile kt+2 stays in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur = cur == 2 ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs of the last two iterations

  const int mode = p.splits > 1 ? 0 : ((p.flags & RNNT_GEMM_ACCUM) ? 2 : 1);
  float* slab = p.splits > 1 ? p.slab + (long)blockIdx.y * p.M * p.N : nullptr;
  float bias_v[4], sb4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + (lane & 15);
    bias_v[j] = (mode != 0 && p.bias && n < p.N) ? p.bias[n] : 0.f;
    sb4[j] = hp_inv_scale_from_amax(p.b_amax[min(n, p.N - 1)]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = m0 + wr * 64 + i * 16 + 4 * (lane >> 4) + reg;
      const bool mok = m < p.M;
      const int mc = mok ? m : 0;
      const float sa = hp_inv_scale_from_amax(p.a_amax[mc]);
      float* crow = mode == 0 ? slab + (long)mc * p.N : p.C + (long)(mc / p.c_div) * p.c_so + (long)(mc % p.c_div) * p.c_si;
      float old[4] = {0.f, 0.f, 0.f, 0.f};
      if (mode == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wc * 64 + j * 16 + (lane & 15);
          if (mok && n < p.N) old[j] = crow[n];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane & 15);
        if (mok && n < p.N) crow[n] = acc[i][j][reg] * sa * sb4[j] + bias_v[j] + old[j];
      }
    }
  }
}

__global__ void __launch_bounds__(256) hp_splitk_reduce_kernel(const HpGemmK p) {
  const long total = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float s = 0.f;
    for (int z = 0; z < p.splits; ++z) s += p.slab[(long)z * total + i];
    if (p.bias) s += p.bias[n];
    const int mo = p.c_rowidx ? p.c_rowidx[m] : m;
    const long off = (long)(mo / p.c_div) * p.c_so + (long)(mo % p.c_div) * p.c_si + n;
    if (p.flags & RNNT_GEMM_ACCUM) s += p.C[off];
    p.C[off] = s;
  }
}

}  // namespace

// internal entry points shared with lstm.hip -------------------------------------------------------------------------
size_t hp_plane_bytes(int64_t rows, int64_t K) { return (size_t)rows * (size_t)ceil_div(K, 32) * 128; }

int hp_colmax(const float* x, int64_t rows, int64_t C, int64_t ld, uint32_t* amax, hipStream_t s) {
  RNNT_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)C * 4, s));
  if (rows == 0 || C == 0) return RNNT_OK;
  long chunks = ceil_div(2048, ceil_div(C, 256));
  if (chunks > ceil_div(rows, 64)) chunks = ceil_div(rows, 64);
  if (chunks < 1) chunks = 1;
  const long rpc = ceil_div(rows, chunks);
  ProfScope prof(RNNT_K_HP_SPLIT, 4.0 * (double)rows * (double)C, s);
  hipLaunchKernelGGL(hp_colmax_kernel, dim3((unsigned)ceil_div(C, 256), (unsigned)ceil_div(rows, rpc)), dim3(256), 0, s, x, (long)rows, (int)C,
                     (long)ld, rpc, amax);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

int hp_split(const float* x, int64_t rows, int64_t K, int64_t ld, uint32_t* amax, void* planes, hipStream_t s, const int* rowidx) {
  if (rows == 0) return RNNT_OK;
  const long blocks = ceil_div(rows, 4);
  ProfScope prof(RNNT_K_HP_SPLIT, 8.0 * (double)rows * (double)K, s);
  hipLaunchKernelGGL(hp_split_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, x, (long)rows, (int)K, (long)ld, amax,
                     (char*)planes, rowidx);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

int hp_split_t(const float* x, int64_t R, int64_t K, int64_t ld, int64_t Ksrc, int64_t shift, const uint32_t* amax, void* planes,
               hipStream_t s, const int* kidx) {
  if (R == 0 || K == 0) return RNNT_OK;
  ProfScope prof(RNNT_K_HP_SPLIT, 8.0 * (double)R * (double)K, s);
  hipLaunchKernelGGL(hp_split_t_kernel, dim3((unsigned)ceil_div(K, 32), (unsigned)ceil_div(R, 256)), dim3(256), 0, s, x, (int)R, (int)K, (long)ld,
                     (int)Ksrc, (int)shift, amax, (char*)planes, kidx);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

int hp_split_both(const float* x, int64_t M, int64_t C, int64_t ld, const uint32_t* rowmax, const uint32_t* colmax, void* planes_rm,
                  void* planes_t, hipStream_t s, const int* rowidx) {
  if (M == 0 || C == 0) return RNNT_OK;
  ProfScope prof(RNNT_K_HP_SPLIT, 12.0 * (double)M * (double)C, s);
  hipLaunchKernelGGL(hp_split_both_kernel, dim3((unsigned)ceil_div(M, 32), (unsigned)ceil_div(C, 256)), dim3(256), 0, s, x, (int)M, (int)C, (long)ld,
                     rowmax, colmax, (char*)planes_rm, (char*)planes_t, rowidx);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

size_t hp_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const long tiles = ceil_div(M, HP_BM) * ceil_div(N, HP_BN);   // the coarser tiling: an upper bound on the splits either kernel takes
  const long nkt = ceil_div(K, HP_BK);
  if (tiles >= 192 || nkt < 64) return 0;
  long want = ceil_div(256, tiles);
  if (want > nkt / 32) want = nkt / 32;
  if (want > 32) want = 32;
  return want >= 2 ? (size_t)want * M * N * 4 : 0;
}

int hp_gemm(const void* A, const uint32_t* a_amax, const void* B, const uint32_t* b_amax, int64_t M, int64_t N, int64_t K, float* C,
            int64_t c_div, int64_t c_so, int64_t c_si, const float* bias, unsigned flags, void* workspace, size_t workspace_bytes,
            hipStream_t s, const int* a_rowidx, int64_t a_plane_rows, const int* c_rowidx) {
  RNNT_CHECK_ARG(A && B && C && a_amax && b_amax, "gemm_hp: null operand");
  RNNT_CHECK_ARG(!a_rowidx || a_plane_rows >= M, "gemm_hp: a gathered A operand needs the row count of its planes");
  RNNT_CHECK_ARG(M >= 1 && N >= 1 && K >= 1 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_hp: bad dims");
  RNNT_CHECK_ARG(c_div >= 1, "gemm_hp: c_div must be >= 1");
  RNNT_CHECK_ARG(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 127) == 0, "gemm_hp: planes must be 128-byte aligned");
  HpGemmK k;
  k.M = (int)M; k.N = (int)N; k.nkt = (int)ceil_div(K, HP_BK);
  const size_t ab = hp_plane_bytes(a_rowidx ? a_plane_rows : M, K), bb = hp_plane_bytes(N, K);
  RNNT_CHECK_ARG(ab < (1ull << 32) && bb < (1ull << 32), "gemm_hp: an operand exceeds the 4 GB a buffer resource addresses");
  k.A = (const char*)A; k.a_pitch = (unsigned)k.nkt * 128u; k.a_bytes = (unsigned)ab;
  k.B = (const char*)B; k.b_pitch = (unsigned)k.nkt * 128u; k.b_bytes = (unsigned)bb;
  k.a_amax = a_amax; k.b_amax = b_amax;
  k.C = C; k.c_div = (int)(c_div > 0x7fffffff ? 0x7fffffff : c_div); k.c_so = c_so; k.c_si = c_si;
  k.bias = bias; k.flags = flags;
  k.a_rowidx = a_rowidx; k.c_rowidx = c_rowidx;
  const bool k3 = getenv("RNNT_GEMM_HP_3STAGE") != nullptr && !a_rowidx && !c_rowidx;   // opt-in: 256x128 tiles / 3-stage LDS ring (measured 10-14 % slower than 256x256 / 2 stages)
  k.tiles_m = (int)ceil_div(M, HP_BM); k.tiles_n = (int)ceil_div(N, k3 ? HP3_BN : HP_BN);
  const int tiles = k.tiles_m * k.tiles_n;
  {  // band height of the tile walk: 8 x 4 tiles in flight per XCD when an XCD's share is >= 32 tiles, 4 x 2 for the small outputs
    static const int env_gm = getenv("RNNT_GEMM_HP_GROUP_M") ? atoi(getenv("RNNT_GEMM_HP_GROUP_M")) : -1;
    k.group_m = env_gm >= 0 ? env_gm : (tiles >= 256 ? 8 : 4);
    if (k3) k.group_m = 0;
  }
  int splits = 1;
  if (workspace && tiles < 192 && k.nkt >= 64) {  // too few tiles for 256 CUs and a deep contraction (weight gradients): split K
    long want = ceil_div(256, tiles);
    const long by_ws = (long)(workspace_bytes / ((size_t)M * N * 4));
    if (want > k.nkt / 32) want = k.nkt / 32;
    if (want > by_ws) want = by_ws;
    if (want > 32) want = 32;
    if (want >= 2) splits = (int)want;
  }
  k.kt_per_split = (int)ceil_div(k.nkt, splits);
  splits = (int)ceil_div(k.nkt, k.kt_per_split);
  k.splits = splits;
  k.slab = (float*)workspace;
  if (k3) RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_hp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HP3_NST * HP3_STAGE));
  else RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_hp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HP_STAGE));
  {
    ProfScope prof(RNNT_K_GEMM_HP, 2.0 * (double)M * (double)N * (double)K, s);
    if (k3) hipLaunchKernelGGL(gemm_hp3_kernel, dim3(tiles, splits), dim3(512), HP3_NST * HP3_STAGE, s, k);
    else hipLaunchKernelGGL(gemm_hp_kernel, dim3(tiles, splits), dim3(512), 2 * HP_STAGE, s, k);
    RNNT_CHECK_LAUNCH();
    if (splits > 1) {
      const long blocks = ceil_div((long)M * N, 256);
      hipLaunchKernelGGL(hp_splitk_reduce_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, k);
      RNNT_CHECK_LAUNCH();
    }
  }
  return RNNT_OK;
}

constexpr int HPQ_SPLIT_CAP = 8;
size_t hp_gemm_grouped_workspace_bytes(const int64_t* MN, int n) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += align_up((size_t)HPQ_SPLIT_CAP * (size_t)MN[i] * 4, 256);
  return total;
}

int hp_gemm_grouped(const HpProblem* pr, int n, unsigned xcd_skip, unsigned* counter, void* workspace, size_t workspace_bytes,
                    hipStream_t s, unsigned* status) {
  RNNT_CHECK_ARG(pr && n >= 1 && n <= HPQ_MAX && counter, "gemm_hp grouped: 1..%d problems and a counter word", HPQ_MAX);
  RNNT_CHECK_ARG((xcd_skip & 0xffu) != 0xffu, "gemm_hp grouped: xcd_skip leaves no XCD");
  HpGemmQ q = {};
  q.nprob = n; q.xcd_skip = xcd_skip; q.counter = counter;
  int xcds = 0;
  for (int x = 0; x < 8; ++x) xcds += !((xcd_skip >> x) & 1u);
  const long workers = 32l * xcds;
  long total_kt = 0;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const HpProblem& a = pr[i];
    RNNT_CHECK_ARG(a.A && a.B && a.C && a.a_amax && a.b_amax, "gemm_hp grouped: null operand (problem %d)", i);
    RNNT_CHECK_ARG(a.M >= 1 && a.N >= 1 && a.K >= 1 && a.M < (1ll << 31) && a.N < (1ll << 31) && a.K < (1ll << 31) && a.ldc >= a.N,
                   "gemm_hp grouped: bad dims (problem %d)", i);
    RNNT_CHECK_ARG(((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 127) == 0, "gemm_hp grouped: planes must be 128-byte aligned");
    HpGemmK& k = q.prob[i];
    k.M = (int)a.M; k.N = (int)a.N; k.nkt = (int)ceil_div(a.K, HP_BK);
    const size_t ab = hp_plane_bytes(a.M, a.K), bb = hp_plane_bytes(a.N, a.K);
    RNNT_CHECK_ARG(ab < (1ull << 32) && bb < (1ull << 32), "gemm_hp grouped: an operand exceeds the 4 GB a buffer resource addresses");
    k.A = (const char*)a.A; k.a_pitch = (unsigned)k.nkt * 128u; k.a_bytes = (unsigned)ab;
    k.B = (const char*)a.B; k.b_pitch = (unsigned)k.nkt * 128u; k.b_bytes = (unsigned)bb;
    k.a_amax = a.a_amax; k.b_amax = a.b_amax;
    k.C = a.C; k.c_div = 1; k.c_so = a.ldc; k.c_si = 0;
    k.bias = nullptr; k.flags = a.flags;
    k.a_rowidx = k.c_rowidx = nullptr;
    k.tiles_m = (int)ceil_div(a.M, HP_BM); k.tiles_n = (int)ceil_div(a.N, HP_BN);
    total_kt += (long)k.tiles_m * k.tiles_n * k.nkt;
    flops += 2.0 * (double)a.M * (double)a.N * (double)a.K;
  }
  // units of (about) equal length, four per resident workgroup: the queue levels whatever imbalance remains
  long chunk = ceil_div(total_kt, 4 * workers);
  if (chunk < 32) chunk = 32;
  size_t off = 0;
  int units = 0;
  for (int i = 0; i < n; ++i) {
    HpGemmK& k = q.prob[i];
    long splits = ceil_div(k.nkt, chunk);
    if (splits > HPQ_SPLIT_CAP) splits = HPQ_SPLIT_CAP;
    const size_t per = (size_t)k.M * k.N * 4;
    while (splits > 1 && (!workspace || off + align_up((size_t)splits * per, 256) > workspace_bytes)) --splits;
    k.kt_per_split = (int)ceil_div(k.nkt, splits);
    k.splits = (int)ceil_div(k.nkt, k.kt_per_split);
    k.slab = k.splits > 1 ? reinterpret_cast<float*>((char*)workspace + off) : nullptr;
    if (k.splits > 1) off += align_up((size_t)k.splits * per, 256);
    units += k.tiles_m * k.tiles_n * k.splits;
    q.unit_end[i] = units;
  }
  RNNT_CHECK_HIP(hipMemsetAsync(counter, 0, 64, s));
  const int lds = 2 * HP_STAGE + 16;
  RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_hpq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  {
    ProfScope prof(RNNT_K_GEMM_HP, flops, s);
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    hipLaunchKernelGGL(gemm_hpq_kernel, dim3(!xcd_skip && units < cus ? units : cus), dim3(512), lds, s, q);   // with skipped XCDs: one per CU, an eighth lands on each XCD
    RNNT_CHECK_LAUNCH();
    hipLaunchKernelGGL(hpq_check_kernel, dim3(1), dim3(1), 0, s, counter, (unsigned)units, status);
    RNNT_CHECK_LAUNCH();
    for (int i = 0; i < n; ++i)
      if (q.prob[i].splits > 1) {
        const long blocks = ceil_div((long)q.prob[i].M * q.prob[i].N, 256);
        hipLaunchKernelGGL(hp_splitk_reduce_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, q.prob[i]);
        RNNT_CHECK_LAUNCH();
      }
  }
  return RNNT_OK;
}

}  // namespace rnnt

using namespace rnnt;

extern "C" size_t rnnt_hip_hp_bytes(int64_t rows, int64_t K) {
  if (rows < 0 || K < 0) return 0;
  return hp_plane_bytes(rows, K);
}

extern "C" int rnnt_hip_hp_split(const float* x, int64_t rows, int64_t K, int64_t ld, int32_t transpose, int64_t src_rows, int64_t shift,
                                 void* planes, uint32_t* amax, int32_t amax_given, void* stream) {
  RNNT_CHECK_ARG(x && planes && amax && rows >= 0 && K >= 0, "hp_split: bad arguments");
  RNNT_CHECK_ARG((reinterpret_cast<uintptr_t>(planes) & 127) == 0, "hp_split: planes must be 128-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!transpose) {
    RNNT_CHECK_ARG(ld >= K, "hp_split: ld < K");
    RNNT_CHECK_ARG(!amax_given, "hp_split: the row-major split computes its row maxima itself");
    return hp_split(x, rows, K, ld, amax, planes, s);
  }
  RNNT_CHECK_ARG(ld >= rows && src_rows >= 1, "hp_split: transposed source is (src_rows x >= rows), ld >= rows");
  if (!amax_given)
    if (int rc = hp_colmax(x, src_rows, rows, ld, amax, s)) return rc;
  return hp_split_t(x, rows, K, ld, src_rows, shift, amax, planes, s);
}

extern "C" size_t rnnt_hip_gemm_hp_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  return hp_gemm_workspace_bytes(M, N, K);
}

extern "C" int rnnt_hip_gemm_hp(const void* A, const uint32_t* a_amax, const void* B, const uint32_t* b_amax, int64_t M, int64_t N,
                                int64_t K, float* C, int64_t ldc, const float* bias, uint32_t flags, void* workspace,
                                size_t workspace_bytes, void* stream) {
  RNNT_CHECK_ARG(ldc >= N, "gemm_hp: ldc < N");
  return hp_gemm(A, a_amax, B, b_amax, M, N, K, C, 1, ldc, 0, bias, flags, workspace, workspace_bytes, (hipStream_t)stream);
}

static_assert(sizeof(rnnt_hp_problem) == sizeof(HpProblem), "rnnt_hp_problem mirrors HpProblem");

extern "C" size_t rnnt_hip_gemm_hp_grouped_workspace_bytes(const rnnt_hp_problem* problems, int32_t n) {
  if (!problems || n < 1 || n > HP_GROUP_MAX) return 0;
  int64_t mn[HP_GROUP_MAX];
  for (int i = 0; i < n; ++i) mn[i] = problems[i].M * problems[i].N;
  return 256 + hp_gemm_grouped_workspace_bytes(mn, n);
}

extern "C" int rnnt_hip_gemm_hp_grouped(const rnnt_hp_problem* problems, int32_t n, uint32_t xcd_skip, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  RNNT_CHECK_ARG(problems && n >= 1 && n <= HP_GROUP_MAX, "gemm_hp grouped: 1..%d problems", HP_GROUP_MAX);
  RNNT_CHECK_ARG(workspace && workspace_bytes >= 256 && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                 "gemm_hp grouped: workspace of >= 256 bytes, 256-byte aligned");
  HpProblem pr[HP_GROUP_MAX];
  for (int i = 0; i < n; ++i) {
    const rnnt_hp_problem& a = problems[i];
    pr[i] = HpProblem{a.A, a.a_amax, a.B, a.b_amax, a.M, a.N, a.K, a.C, a.ldc, a.flags};
  }
  return hp_gemm_grouped(pr, n, xcd_skip, reinterpret_cast<unsigned*>(workspace), (char*)workspace + 256, workspace_bytes - 256,
                         (hipStream_t)stream);
}

extern "C" int rnnt_hip_hp_split_both(const float* x, int64_t M, int64_t C, int64_t ld, const uint32_t* rowmax, const uint32_t* colmax,
                                      void* planes_rm, void* planes_t, void* stream) {
  RNNT_CHECK_ARG(x && rowmax && colmax && planes_rm && planes_t && M >= 0 && C >= 0 && ld >= C, "hp_split_both: bad arguments");
  RNNT_CHECK_ARG(M < (1ll << 31) && C < (1ll << 31), "hp_split_both: dims must fit 31 bits");
  RNNT_CHECK_ARG(((reinterpret_cast<uintptr_t>(planes_rm) | reinterpret_cast<uintptr_t>(planes_t)) & 127) == 0, "hp_split_both: planes must be 128-byte aligned");
  return hp_split_both(x, M, C, ld, rowmax, colmax, planes_rm, planes_t, (hipStream_t)stream);
}
