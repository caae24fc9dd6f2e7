// fp32 GEMM on the f32-input matrix cores of gfx950 (v_mfma_f32_32x32x2_f32; exact fp32 fma chains).
//
// Replaces the BLAS calls behind nn.Linear (networks/encoder.py:76,103; networks/decoder.py:80,124;
// networks/transducer.py:39,69) and the hoisted W_ih.x_t input projection of nn.LSTM
// (networks/encoder.py:67-75,99), plus their backward GEMMs.  See include/rnnt_hip.h for operand maps.
//
// Tile: 128x128x16 per 256-thread workgroup (4 waves, one per SIMD, 64x64 per wave = 2x2 MFMA tiles),
// operands staged K-major in LDS (row stride 132 floats: ds_read_b32 of 32 consecutive m/n is
// conflict-free, ds_write_b128 stays 16-B aligned), register-prefetch double buffering.
#include "common.hpp"

#include <stdlib.h>

namespace rnnt {

namespace {

constexpr int BM = 128, BK = 16, LDT = 132;

struct GemmK {
  int M, N, K;
  const float* A;
  int a_div;
  long a_so, a_si, a_sk;
  const long* a_rowidx;
  const float* B;
  long b_sn, b_sk;
  float* C;
  int c_div;
  long c_so, c_si;
  const float* bias;
  const float* aux;
  unsigned flags;
  int splits, kchunk;  // split-K: blockIdx.y = split z handles k in [z*kchunk, min(K, (z+1)*kchunk))
  float* slab;         // splits > 1: partial products go to slab[z][M][N], reduced by splitk_reduce_kernel
};

// Loads this thread's share (2 x 4 floats) of a (128 rows) x (16 k) operand tile and parks it in LDS K-major.
// KC: operand rows are k-contiguous in memory (row r at rowptr[r], element k at +k).
// !KC: operand is r-contiguous (element (r,k) at base + k*sk + r).
template <bool KC, bool VEC4, int ROWS>
struct TileIO {
  static constexpr int NP = ROWS / 64;          // float4 per thread per K-tile
  static constexpr int LD = ROWS + 4;           // LDS row stride (floats) of the K-major image
  static constexpr int TPR = ROWS / 4;          // !KC: threads per k-row
  static constexpr int KPP = 256 / TPR;         // !KC: k-rows per pass
  f32x4 v[NP];
  // KC state
  const float* rowptr[NP];
  // !KC state
  const float* base;
  long sk;
  int r0, R, K;
  bool gelu;

  __device__ __forceinline__ void load(int k0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if constexpr (KC) {
        const int k = k0 + 4 * (tid & 3);
        const float* src = rowptr[p];
        if (src != nullptr) {
          if (VEC4 && k + 3 < K) {
            x = *reinterpret_cast<const f32x4*>(src + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < K) x[e] = src[k + e];
          }
        }
      } else {
        const int k = k0 + tid / TPR + KPP * p;
        const int r = r0 + 4 * (tid % TPR);
        if (k < K) {
          const float* src = base + (long)k * sk + r;
          if (VEC4 && r + 3 < R) {
            x = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r + e < R) x[e] = src[e];
          }
        }
      }
      if (gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = gelu_tanh(x[e]);
      }
      v[p] = x;
    }
  }

  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if constexpr (KC) {
        const int rl = (tid >> 2) + 64 * p;
        const int kq = 4 * (tid & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(kq + e) * LD + rl] = v[p][e];
      } else {
        const int kl = tid / TPR + KPP * p;
        *reinterpret_cast<f32x4*>(&S[kl * LD + 4 * (tid % TPR)]) = v[p];
      }
    }
  }
};

// operand-row pointers / bases of this workgroup's A and B tiles (shared by the fp32 and the split-bf16 kernels)
template <bool A_KC, bool B_KC, int BN, typename TA, typename TB>
__device__ __forceinline__ void setup_io(TA& ta, TB& tb, const GemmK& p, int m0, int n0) {
  const int tid = threadIdx.x;
  ta.K = tb.K = p.K;
  ta.gelu = (p.flags & RNNT_GEMM_GELU_A) != 0;
  tb.gelu = (p.flags & RNNT_GEMM_GELU_B) != 0;
  if constexpr (A_KC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = m0 + (tid >> 2) + 64 * q;
      if (m < p.M) {
        long off = p.a_rowidx ? p.a_rowidx[m] * p.a_si : (long)(m / p.a_div) * p.a_so + (long)(m % p.a_div) * p.a_si;
        ta.rowptr[q] = p.A + off;
      } else {
        ta.rowptr[q] = nullptr;
      }
    }
  } else {
    ta.base = p.A;
    ta.sk = p.a_sk;
    ta.r0 = m0;
    ta.R = p.M;
  }
  if constexpr (B_KC) {
#pragma unroll
    for (int q = 0; q < BN / 64; ++q) {
      const int n = n0 + (tid >> 2) + 64 * q;
      tb.rowptr[q] = n < p.N ? p.B + (long)n * p.b_sn : nullptr;
    }
  } else {
    tb.base = p.B;
    tb.sk = p.b_sk;
    tb.r0 = n0;
    tb.R = p.N;
  }
}

// C/D map of the 32x32 MFMAs (f32 and bf16 forms alike): col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
template <int NW>
__device__ __forceinline__ void epilogue(const GemmK& p, const f32x16 (&acc)[2][NW], int m0, int n0, int wm, int wn, int lane) {
  if (p.splits > 1) {
    float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int n = n0 + wn * (32 * NW) + j * 32 + (lane & 31);
          if (n < p.N) slab[(long)m * p.N + n] = acc[i][j][v];
        }
      }
    return;
  }
  const bool accum = (p.flags & RNNT_GEMM_ACCUM) != 0, dgelu = (p.flags & RNNT_GEMM_MUL_DGELU) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
      if (m >= p.M) continue;
      const long rowoff = (long)(m / p.c_div) * p.c_so + (long)(m % p.c_div) * p.c_si;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int n = n0 + wn * (32 * NW) + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        float val = acc[i][j][v];
        if (p.bias) val += p.bias[n];
        const long off = rowoff + n;
        if (dgelu) val *= dgelu_tanh(p.aux[off]);
        if (accum) val += p.C[off];
        p.C[off] = val;
      }
    }
  }
}

// bijective regrouping of block ids so that the tiles one XCD works on are neighbours (ids are dealt round-robin
// over the 8 XCDs); consecutive regrouped ids walk down M first
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// NW = 32-wide N tiles per wave: 2 -> 128x128 block tile, 4 -> 128x256 (each wave 64x128: 8 MFMAs per operand fetch,
// 25% fewer operand bytes per FLOP; used when N is wide enough)
template <bool A_KC, bool B_KC, bool VEC4, int NW>
__global__ void __launch_bounds__(256, (NW == 4 ? 2 : 3)) gemm_f32_kernel(const GemmK p) {
  constexpr int BN = 64 * NW, LDB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // tile coordinates: consecutive block ids walk down M first so that the (usually small) B panel and a
  // band of A stay in the XCD's L2; ids are dealt round-robin over 8 XCDs, so regroup them (bijective map).
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;

  TileIO<A_KC, VEC4, BM> ta;
  TileIO<B_KC, VEC4, BN> tb;
  setup_io<A_KC, B_KC, BN>(ta, tb, p, m0, n0);

  f32x16 acc[2][NW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  ta.K = tb.K = kend;
  const int nk = (kend - kbeg + BK - 1) / BK;
  ta.load(kbeg);
  tb.load(kbeg);
  ta.store(As[0]);
  tb.store(Bs[0]);
  __syncthreads();

  const int aoff = (lane >> 5) * LDT + wm * 64 + (lane & 31);
  const int boff = (lane >> 5) * LDB + wn * (32 * NW) + (lane & 31);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    ta.load(kbeg + (kt + 1) * BK);  // past the last tile every element is predicated off (k >= kend): no access, zeros
    tb.load(kbeg + (kt + 1) * BK);
    const float* as = As[cur] + aoff;
    const float* bs = Bs[cur] + boff;
    // all of this K-tile's operands first (32 VGPRs), then 32 MFMAs back to back: the LDS latency is exposed once per
    // K-tile instead of once per 4 MFMAs (the other waves of the SIMD cover that one)
    // Two batches of 4 k-pairs.  Batch 0: operand reads, then its MFMAs (the next tile's global loads, issued above,
    // land meanwhile).  Batch 1: operand reads, then its MFMAs with the next tile's LDS stores slotted between them
    // (other LDS buffer, no hazard): the vmcnt wait and the store pass ride in the MFMA shadow instead of forming
    // their own phase in front of the barrier.
    constexpr int KH = 4;
#pragma unroll
    for (int kb = 0; kb < BK / 2; kb += KH) {
      float av[KH][2], bv[KH][NW];
#pragma unroll
      for (int kp = 0; kp < KH; ++kp) {
        av[kp][0] = as[2 * (kb + kp) * LDT];
        av[kp][1] = as[2 * (kb + kp) * LDT + 32];
#pragma unroll
        for (int j = 0; j < NW; ++j) bv[kp][j] = bs[2 * (kb + kp) * LDB + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kp = 0; kp < KH; ++kp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp][i], bv[kp][j], acc[i][j], 0, 0, 0);
      if (kb + KH >= BK / 2) {
        ta.store(As[cur ^ 1]);  // unconditional (straight-line body): after the last tile this parks zeros/stale data in
        tb.store(Bs[cur ^ 1]);  // the buffer nobody reads again
        // schedule: 2 MFMAs, then one DS write, repeated; the remaining MFMAs follow
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  epilogue<NW>(p, acc, m0, n0, wm, wn, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores by operand splitting.
//
// x = x0 + x1 + x2 EXACTLY, each piece 8 significant bits (truncation split: x0 = x & 0xffff0000, r = x - x0 is exact,
// x1 = r & 0xffff0000, x2 = r - x1 has <= 8 bits left), so every piece is a bf16 and every piece product is exact in the
// fp32 accumulator.  a.b = sum_{i+j<=2} a_i b_j + O(2^-24 |a||b|): six v_mfma_f32_32x32x16_bf16 per 32x32x16 block
// (192 MFMA cycles) against eight v_mfma_f32_32x32x2_f32 (512): the same fp32 inputs, outputs, accumulators and
// error order as the exact-fp32 kernel above at 2.67x its matrix-core rate.  NPROD = 3 keeps only i+j<=1 (error 2^-16,
// round-to-nearest second piece): opt-in.
//
// LDS: one [rows][16 k] bf16 image per piece; row r at byte (r&3)*X + (r>>2)*48 with X = 64 (mod 256): the ds_read_b128
// operand fetch (32 consecutive rows per half-wave) is conflict-free, and both producers - a thread holding 4 k of one
// row (k-contiguous operand) or 2-4 k of 4 neighbouring rows (row-contiguous operand) - store with at most 2-4-way
// conflicts, which ds_write_b32/b64 absorb in their issue time.
// ------------------------------------------------------------------------------------------------------------------
template <int ROWS, int BKS>
struct PlaneImg {
  static constexpr int RS = 2 * BKS + 16;           // bytes per row: payload + 16 pad (odd number of 16-B slots)
  static constexpr int X = (ROWS / 4) * RS + 64;    // distance between the four row classes r&3
  static constexpr int BYTES = 4 * X;
  static_assert(((ROWS / 4) * RS) % 256 == 0, "class distance must be 64 (mod 256)");
  static __device__ __forceinline__ int off(int r) { return (r & 3) * X + (r >> 2) * RS; }
};

// Global fp32 -> registers -> NPL bf16 piece images in LDS, for a (ROWS) x (BKS k) operand tile and NT threads.
// KC (k-contiguous rows): BKS/4 threads per row, each one float4 = 4 consecutive k; with BKS = 32 a row's 128 bytes are
// one whole cache line.  !KC (row-contiguous): a thread takes 4 neighbouring rows x NP CONSECUTIVE k.
template <bool KC, bool VEC4, int ROWS, int BKS, int NPL, int NT = 256>
struct SplitIO {
  static constexpr int NP = ROWS * BKS / (4 * NT);  // float4 per thread per K-tile
  static constexpr int TPK = BKS / 4;               // KC: threads per row
  static constexpr int RPP = NT / TPK;              // KC: rows per pass
  static constexpr int TPR = ROWS / 4;              // !KC: threads per k-row
  static_assert(NP == 2 || NP == 4, "tile shape");
  using Img = PlaneImg<ROWS, BKS>;
  f32x4 v[NP];
  const float* rowptr[NP];
  const float* base;
  long sk;
  int r0, R, K;
  bool gelu;

  __device__ __forceinline__ void load(int k0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if constexpr (KC) {
        const int k = k0 + 4 * (tid % TPK);
        const float* src = rowptr[p];
        if (src != nullptr) {
          if (VEC4 && k + 3 < K) {
            x = *reinterpret_cast<const f32x4*>(src + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < K) x[e] = src[k + e];
          }
        }
      } else {
        const int k = k0 + (tid / TPR) * NP + p;
        const int r = r0 + 4 * (tid % TPR);
        if (k < K) {
          const float* src = base + (long)k * sk + r;
          if (VEC4 && r + 3 < R) {
            x = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r + e < R) x[e] = src[e];
          }
        }
      }
      if (gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = gelu_tanh(x[e]);
      }
      v[p] = x;
    }
  }

  __device__ __forceinline__ void store(char* S) const {
    const int tid = threadIdx.x;
    if constexpr (KC) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        unsigned lo[NPL], hi[NPL];
        split_pair<NPL>(v[p][0], v[p][1], lo);
        split_pair<NPL>(v[p][2], v[p][3], hi);
        char* dst = S + Img::off(tid / TPK + RPP * p) + 8 * (tid % TPK);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x2*>(dst + pl * Img::BYTES) = u32x2{lo[pl], hi[pl]};
      }
    } else {
      const int kb = (tid / TPR) * NP;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        char* dst = S + Img::off(4 * (tid % TPR) + e) + 2 * kb;
        unsigned lo[NPL];
        split_pair<NPL>(v[0][e], v[1][e], lo);
        if constexpr (NP == 2) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<unsigned*>(dst + pl * Img::BYTES) = lo[pl];
        } else {
          unsigned hi[NPL];
          split_pair<NPL>(v[2][e], v[3][e], hi);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x2*>(dst + pl * Img::BYTES) = u32x2{lo[pl], hi[pl]};
        }
      }
    }
  }
};

// row pointers of the split kernels' k-contiguous operands (RPP rows per pass instead of 64)
template <bool A_KC, bool B_KC, typename TA, typename TB>
__device__ __forceinline__ void setup_split_io(TA& ta, TB& tb, const GemmK& p, int m0, int n0) {
  const int tid = threadIdx.x;
  ta.K = tb.K = p.K;
  ta.gelu = (p.flags & RNNT_GEMM_GELU_A) != 0;
  tb.gelu = (p.flags & RNNT_GEMM_GELU_B) != 0;
  if constexpr (A_KC) {
#pragma unroll
    for (int q = 0; q < TA::NP; ++q) {
      const int m = m0 + tid / TA::TPK + TA::RPP * q;
      if (m < p.M) {
        long off = p.a_rowidx ? p.a_rowidx[m] * p.a_si : (long)(m / p.a_div) * p.a_so + (long)(m % p.a_div) * p.a_si;
        ta.rowptr[q] = p.A + off;
      } else {
        ta.rowptr[q] = nullptr;
      }
    }
  } else {
    ta.base = p.A;
    ta.sk = p.a_sk;
    ta.r0 = m0;
    ta.R = p.M;
  }
  if constexpr (B_KC) {
#pragma unroll
    for (int q = 0; q < TB::NP; ++q) {
      const int n = n0 + tid / TB::TPK + TB::RPP * q;
      tb.rowptr[q] = n < p.N ? p.B + (long)n * p.b_sn : nullptr;
    }
  } else {
    tb.base = p.B;
    tb.sk = p.b_sk;
    tb.r0 = n0;
    tb.R = p.N;
  }
}

template <bool A_KC, bool B_KC, bool VEC4, int NW, int NPROD, int BKS, bool DBG = false>
__global__ void __launch_bounds__(256, 2) gemm_bf16s_kernel(const GemmK p) {
  constexpr int BN = 64 * NW;
  constexpr int NPL = NPROD == 6 ? 3 : 2;
  using IA = PlaneImg<BM, BKS>;
  using IB = PlaneImg<BN, BKS>;
  __shared__ __attribute__((aligned(16))) char As[NPL * IA::BYTES];
  __shared__ __attribute__((aligned(16))) char Bs[NPL * IB::BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;

  SplitIO<A_KC, VEC4, BM, BKS, NPL> ta;
  SplitIO<B_KC, VEC4, BN, BKS, NPL> tb;
  setup_split_io<A_KC, B_KC>(ta, tb, p, m0, n0);

  f32x16 acc[2][NW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  ta.K = tb.K = kend;
  const int nk = (kend - kbeg + BKS - 1) / BKS;
  ta.load(kbeg);
  tb.load(kbeg);

  // MFMA operand fetch: lane -> row (lane&31) of a 32-row block, k 8*(lane>>5) .. +7 (16 bytes) of a 16-k step
  const char* ard[2];
  const char* brd[NW];
#pragma unroll
  for (int i = 0; i < 2; ++i) ard[i] = As + IA::off(wm * 64 + i * 32 + (lane & 31)) + 16 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < NW; ++j) brd[j] = Bs + IB::off(wn * (32 * NW) + j * 32 + (lane & 31)) + 16 * (lane >> 5);

  long ph[6] = {0, 0, 0, 0, 0, 0};  // DBG: cycles in barrier-1 | vmcnt wait | split+store | barrier-2 | load issue | ds_read+MFMA
  for (int kt = 0; kt < nk; ++kt) {
    long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (DBG) c0 = clock64();
    if (kt) __syncthreads();  // every wave is done reading the previous K-tile's images
    if (DBG) {
      c1 = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      c2 = clock64();
    }
    ta.store(As);
    tb.store(Bs);
    if (DBG) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      c3 = clock64();
    }
    __syncthreads();
    if (DBG) c4 = clock64();
    ta.load(kbeg + (kt + 1) * BKS);  // lands behind the MFMAs; past the last tile everything is predicated off
    tb.load(kbeg + (kt + 1) * BKS);
    if (DBG) {
      c5 = clock64();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < BKS / 16; ++ks) {
      bf16x8 af[2][NPL], bf[NW][NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ard[i] + pl * IA::BYTES + 32 * ks));
#pragma unroll
        for (int j = 0; j < NW; ++j)
          bf[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(brd[j] + pl * IB::BYTES + 32 * ks));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          // smallest terms first
          if constexpr (NPROD == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
        }
    }
    if (DBG) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][NW - 1][15]));  // the last MFMAs have retired
      const long c6 = clock64();
      ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4; ph[5] += c6 - c5;
    }
  }
  if (DBG && lane == 0 && p.slab != nullptr && blockIdx.x % 64 == 0 && blockIdx.x / 64 < 32) {
    long* out = reinterpret_cast<long*>(p.slab) + ((blockIdx.x / 64) * 4 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] = ph[i];
    out[6] = nk;
  }
  epilogue<NW>(p, acc, m0, n0, wm, wn, lane);
}

// 256x256x16 tile, 512 threads (8 waves as 4 x 2, each 64 x 128), ONE workgroup per CU, bf16 piece images double-buffered
// in 147 KB of LDS, one barrier per K-tile.  Why (tools/gemm_phase_probe.py on the 128x256 form): a CU's vector-memory path
// needs 1.5-3.3 k cycles to take the six wave-loads of a K-tile from each of its 8 waves, and nothing overlapped that with
// the 1.5 k cycles of MFMAs; this form moves a third fewer operand bytes per FLOP through that path (4 wave-loads per
// 64x128x16 of output instead of 6) and staggers the two waves of every SIMD: waves 0-3 split + store the next tile and issue
// their loads BEFORE their MFMAs, waves 4-7 AFTER, so one wave's VALU / LDS-store / load-issue phase runs under the other
// wave's MFMAs.
template <bool A_KC, bool B_KC, bool VEC4, int NPROD>
__global__ void __launch_bounds__(512, 1) gemm_bf16s256_kernel(const GemmK p) {
  constexpr int BMB = 256, BNB = 256, NW = 4, NT = 512;
  constexpr int NPL = NPROD == 6 ? 3 : 2;
  using IA = PlaneImg<BMB, BK>;
  using IB = PlaneImg<BNB, BK>;
  constexpr int STAGE = NPL * (IA::BYTES + IB::BYTES);
  extern __shared__ __attribute__((aligned(16))) char lds256[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (p.M + BMB - 1) / BMB, tiles_n = (p.N + BNB - 1) / BNB;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (bid % tiles_m) * BMB, n0 = (bid / tiles_m) * BNB;

  SplitIO<A_KC, VEC4, BMB, BK, NPL, NT> ta;
  SplitIO<B_KC, VEC4, BNB, BK, NPL, NT> tb;
  setup_split_io<A_KC, B_KC>(ta, tb, p, m0, n0);

  f32x16 acc[2][NW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  ta.K = tb.K = kend;
  const int nk = (kend - kbeg + BK - 1) / BK;

  int aoff[2], boff[NW];
#pragma unroll
  for (int i = 0; i < 2; ++i) aoff[i] = IA::off(wm * 64 + i * 32 + (lane & 31)) + 16 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < NW; ++j) boff[j] = NPL * IA::BYTES + IB::off(wn * 128 + j * 32 + (lane & 31)) + 16 * (lane >> 5);

  auto stage_store = [&](int buf) {
    ta.store(lds256 + buf * STAGE);
    tb.store(lds256 + buf * STAGE + NPL * IA::BYTES);
  };
  auto mma_tile = [&](int buf) {
    const char* base = lds256 + buf * STAGE;
    bf16x8 af[2][NPL], bf[NW][NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + aoff[i] + pl * IA::BYTES));
#pragma unroll
      for (int j = 0; j < NW; ++j) bf[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + boff[j] + pl * IB::BYTES));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if constexpr (NPROD == 6) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };

  ta.load(kbeg);
  tb.load(kbeg);
  stage_store(0);
  ta.load(kbeg + BK);
  tb.load(kbeg + BK);
  __syncthreads();
  const bool early = wave < 4;  // wave-uniform: which half of the workgroup feeds the next tile before its MFMAs
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (early) {
      stage_store(cur ^ 1);                 // tile kt+1 (zeros past the end); buffer cur^1 was last read before the previous barrier
      ta.load(kbeg + (kt + 2) * BK);
      tb.load(kbeg + (kt + 2) * BK);
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(cur);
    } else {
      mma_tile(cur);
      __builtin_amdgcn_sched_barrier(0);
      stage_store(cur ^ 1);
      ta.load(kbeg + (kt + 2) * BK);
      tb.load(kbeg + (kt + 2) * BK);
    }
    __syncthreads();
  }
  epilogue<NW>(p, acc, m0, n0, wm, wn, lane);
}

// how the GEMMs multiply: 6 = split-bf16 with all terms of fp32 weight (default), 3 = split-bf16 first-order only,
// 0 = v_mfma_f32_32x32x2_f32 (exact fp32 fma chains).  RNNT_GEMM_MODE = bf16x6 | bf16x3 | f32.
inline int gemm_mode() {
  const char* e = getenv("RNNT_GEMM_MODE");
  if (e == nullptr || e[0] == 0) return 6;
  if (e[0] == 'f') return 0;
  return (e[0] == 'b' && e[4] == 'x' && e[5] == '3') ? 3 : 6;
}

// fixed-order sum of the split-K slabs + the epilogue the single-pass kernel would have applied
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmK p) {
  const long total = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float s = 0.f;
    for (int z = 0; z < p.splits; ++z) s += p.slab[(long)z * total + i];
    if (p.bias) s += p.bias[n];
    const long off = (long)(m / p.c_div) * p.c_so + (long)(m % p.c_div) * p.c_si + n;
    if (p.flags & RNNT_GEMM_MUL_DGELU) s *= dgelu_tanh(p.aux[off]);
    if (p.flags & RNNT_GEMM_ACCUM) s += p.C[off];
    p.C[off] = s;
  }
}

// 128x256 tiles when N is wide and the grid still has plenty of tiles; 128x128 otherwise
inline int pick_bn(int64_t M, int64_t N) {
  if (getenv("RNNT_GEMM_BN128")) return 128;
  return (N >= 256 && N % 256 == 0 && ceil_div(M, BM) * (N / 256) >= 64) ? 256 : 128;
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

}  // namespace

}  // namespace rnnt

extern "C" size_t rnnt_hip_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  // enough for the split count rnnt_hip_gemm_f32 would pick; 0 when it would not split
  if (M <= 0 || N <= 0 || K < 8 * rnnt::BK) return 0;
  // the larger of what the 128-row tilings (2 workgroups per CU) and the 256x256 tiling (1 per CU) would ask for
  const long by_k = K / (8 * rnnt::BK);
  long best = 0;
  for (int big = 0; big < 2; ++big) {
    const long tiles = big ? rnnt::ceil_div(M, 256) * rnnt::ceil_div(N, 256)
                           : rnnt::ceil_div(M, rnnt::BM) * rnnt::ceil_div(N, 128);
    if (tiles >= (big ? 256 : 512)) continue;
    long want = rnnt::ceil_div(big ? 512 : 1024, tiles);
    if (want > by_k) want = by_k;
    if (want > 64) want = 64;
    if (want > best) best = want;
  }
  return best >= 2 ? (size_t)best * M * N * 4 : 0;
}

extern "C" int rnnt_hip_gemm_f32(const rnnt_gemm_desc* d, void* stream) {
  using namespace rnnt;
  RNNT_CHECK_ARG(d != nullptr, "gemm: null descriptor");
  RNNT_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "gemm: negative dims");
  RNNT_CHECK_ARG(d->M < (1ll << 31) && d->N < (1ll << 31) && d->K < (1ll << 31), "gemm: dims must fit int32");
  if (d->M == 0 || d->N == 0) return RNNT_OK;
  RNNT_CHECK_ARG(d->A && d->B && d->C, "gemm: null operand");
  RNNT_CHECK_ARG(d->c_div >= 1, "gemm: c_div must be >= 1");
  RNNT_CHECK_ARG(!(d->flags & RNNT_GEMM_MUL_DGELU) || d->aux, "gemm: MUL_DGELU needs aux");
  const bool b_kc = d->b_sk == 1;
  RNNT_CHECK_ARG(b_kc || d->b_sn == 1, "gemm: B must be contiguous along k or along n");
  const bool a_kc = d->a_mc == 0;
  if (a_kc) {
    RNNT_CHECK_ARG(d->a_sk == 1, "gemm: k-contiguous A needs a_sk == 1");
    RNNT_CHECK_ARG(d->a_rowidx || d->a_div >= 1, "gemm: a_div must be >= 1");
  } else {
    RNNT_CHECK_ARG(d->a_rowidx == nullptr, "gemm: a_rowidx only with k-contiguous A");
  }

  GemmK k;
  k.M = (int)d->M; k.N = (int)d->N; k.K = (int)d->K;
  k.A = d->A; k.a_div = (int)(d->a_div < 1 ? 1 : (d->a_div > 0x7fffffff ? 0x7fffffff : d->a_div));
  k.a_so = d->a_so; k.a_si = d->a_si; k.a_sk = d->a_sk; k.a_rowidx = (const long*)d->a_rowidx;
  k.B = d->B; k.b_sn = d->b_sn; k.b_sk = d->b_sk;
  k.C = d->C; k.c_div = (int)(d->c_div > 0x7fffffff ? 0x7fffffff : d->c_div); k.c_so = d->c_so; k.c_si = d->c_si;
  k.bias = d->bias; k.aux = d->aux; k.flags = d->flags;

  bool vec = aligned16(d->A) && aligned16(d->B);
  if (a_kc) vec = vec && (d->a_si % 4 == 0) && (d->a_rowidx || d->a_so % 4 == 0);
  else vec = vec && (d->a_sk % 4 == 0);
  if (b_kc) vec = vec && (d->b_sn % 4 == 0);
  else vec = vec && (d->b_sk % 4 == 0);

  const int mode = (d->flags & RNNT_GEMM_EXACT_F32) ? 0 : gemm_mode();
  const int bks = BK;  // (a K-tile depth of 32 with 128x128 tiles was measured slower: DESIGN.md 4.4; the kernel template still takes it)
  const int bn = pick_bn(d->M, d->N);
  // 256x256 tiles / 512 threads / one workgroup per CU for the split-bf16 form when the output is large in both directions
  // (a 256x256 tile walks its K-tiles alone: with fewer than 64 workgroups even after split-K — the prediction net's 1312-row
  //  products — the 128-row tilings finish sooner: 112 -> 30 us for 1312 x 512 x 512)
  bool big = mode == 6 && d->M >= 256 && d->N >= 256 && !getenv("RNNT_GEMM_NO256");
  if (big && !getenv("RNNT_GEMM_256_ALWAYS")) {
    const long t256 = ceil_div(d->M, 256) * ceil_div(d->N, 256);
    long sp = 1;
    if (d->workspace && d->K >= 8 * BK) {
      sp = d->K / (8 * BK);
      const long by_ws = (long)(d->workspace_bytes / ((size_t)d->M * d->N * 4));
      if (sp > by_ws) sp = by_ws;
      if (sp > 64) sp = 64;
      if (sp < 1) sp = 1;
    }
    if (t256 * sp < 64) big = false;
  }
  const int tiles = big ? (int)(ceil_div(d->M, 256) * ceil_div(d->N, 256)) : (int)(ceil_div(d->M, BM) * ceil_div(d->N, bn));
  const int slots = big ? 256 : 512;  // workgroups the chip holds at once
  // split-K when the output has too few tiles to fill 256 CUs and K is deep (weight-gradient GEMMs):
  // partial slabs in the caller's workspace, summed in fixed order (bitwise reproducible; no float atomics)
  int splits = 1;
  if (d->workspace && tiles < slots && d->K >= 8 * BK) {
    long want = ceil_div(2 * slots, tiles);
    const long by_k = d->K / (8 * BK);
    const long by_ws = (long)(d->workspace_bytes / ((size_t)d->M * d->N * 4));
    if (want > by_k) want = by_k;
    if (want > by_ws) want = by_ws;
    if (want > 64) want = 64;
    if (want >= 2) splits = (int)want;
  }
  k.kchunk = splits > 1 ? (int)(ceil_div(ceil_div(d->K, splits), bks) * bks) : (int)(d->K > 0 ? d->K : 1);
  if (splits > 1) splits = (int)ceil_div(d->K, k.kchunk);
  k.splits = splits;
  k.slab = (float*)d->workspace;
  dim3 grid(tiles, splits), block(256);
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS256_6 = 2 * 3 * (PlaneImg<256, BK>::BYTES * 2);
  ProfScope prof(RNNT_K_GEMM, 2.0 * (double)d->M * (double)d->N * (double)d->K, s);
#define LAUNCH_BIG(AK, BKC, V)                                                                                          \
  do {                                                                                                                  \
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16s256_kernel<AK, BKC, V, 6>,                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_6));                          \
    hipLaunchKernelGGL((gemm_bf16s256_kernel<AK, BKC, V, 6>), grid, dim3(512), LDS256_6, s, k);                         \
  } while (0)
#define LAUNCH_K(KERNEL4, KERNEL2)                                        \
  do {                                                                    \
    if (bn == 256) hipLaunchKernelGGL((KERNEL4), grid, block, 0, s, k);   \
    else hipLaunchKernelGGL((KERNEL2), grid, block, 0, s, k);             \
  } while (0)
#define LAUNCH(AK, BKC, V)                                                                                              \
  do {                                                                                                                  \
    if (big) LAUNCH_BIG(AK, BKC, V);                                                                                    \
    else if (mode == 6 && bks == 16 && bn == 256 && splits == 1 && d->workspace && getenv("RNNT_GEMM_DBG"))             \
      hipLaunchKernelGGL((gemm_bf16s_kernel<AK, BKC, V, 4, 6, 16, true>), grid, block, 0, s, k);                        \
    else if (mode == 6) LAUNCH_K((gemm_bf16s_kernel<AK, BKC, V, 4, 6, 16>), (gemm_bf16s_kernel<AK, BKC, V, 2, 6, 16>)); \
    else if (mode == 3) LAUNCH_K((gemm_bf16s_kernel<AK, BKC, V, 4, 3, 16>), (gemm_bf16s_kernel<AK, BKC, V, 2, 3, 16>)); \
    else LAUNCH_K((gemm_f32_kernel<AK, BKC, V, 4>), (gemm_f32_kernel<AK, BKC, V, 2>));                                  \
  } while (0)
  if (a_kc && b_kc) { if (vec) LAUNCH(true, true, true); else LAUNCH(true, true, false); }
  else if (a_kc && !b_kc) { if (vec) LAUNCH(true, false, true); else LAUNCH(true, false, false); }
  else if (!a_kc && b_kc) { if (vec) LAUNCH(false, true, true); else LAUNCH(false, true, false); }
  else { if (vec) LAUNCH(false, false, true); else LAUNCH(false, false, false); }
#undef LAUNCH_K
#undef LAUNCH
#undef LAUNCH_BIG
  RNNT_CHECK_LAUNCH();
  if (splits > 1) {
    const long blocks = ceil_div((long)d->M * d->N, 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), block, 0, s, k);
    RNNT_CHECK_LAUNCH();
  }
  return RNNT_OK;
}
