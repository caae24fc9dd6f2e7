// Fused joint + log-softmax + RNN-T lattice for gfx950.
//
// Replaces JointNet.joint (networks/transducer.py:54-69: repeat/cat/GELU/Linear materialising
// (B,T,U+1,2*O) three times and (B,T,U+1,V) once) followed by RNNTLoss(blank, reduction="mean")
// (model.py:39,57; arithmetic in third-party warp-transducer / torchaudio).  Uses the separable form
// z[b,t,u,:] = A[b,t,:] + C[b,u,:] + bias so only two floats per lattice cell are ever
// written: log p(blank) and log p(y_u).
//
// Kernels (all HBM/latency-bound integer+transcendental work; no MFMA here by design):
//   lse_sep_kernel / lse_dense_kernel : per-cell log-sum-exp -> blk[b][u][t], emit[b][u][t] (u-major)
//   alphabeta_kernel<K>               : one wavefront per (utterance, alpha|beta); lane l owns label
//                                       positions [l*K, l*K+K) and sweeps t with a one-lane skew, so each
//                                       step advances one anti-diagonal; neighbour hand-off by lane shuffle;
//                                       accumulators in fp64 with an fp32 log1p(exp(.)) correction term
//   grad_sep_kernel                   : dA[b,t,:] = sum_u dz, partial dC slabs per 32-frame tile (LDS adds)
//   reduce_dc_kernel                  : fixed-order sum of the slabs (bitwise reproducible, no float atomics)
//   grad_dense_kernel                 : warp-transducer-shaped d/d logits for rnnt_hip_loss_from_logits_*
#include "common.hpp"

#include <stdlib.h>

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace rnnt {
namespace {

constexpr int TT = 32;  // frames per workgroup tile

// dense-logits entry points accept fp32, fp16 or bf16 storage (torchaudio's RNNTLoss takes half logits, model.py:28-31);
// all arithmetic stays fp32 / fp64
__device__ __forceinline__ float ldf(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float ldf(const __half* p, long i) { return __half2float(p[i]); }
__device__ __forceinline__ float ldf(const __hip_bfloat16* p, long i) { return __bfloat162float(p[i]); }
__device__ __forceinline__ void stf(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(__half* p, long i, float v) { p[i] = __float2half(v); }
__device__ __forceinline__ void stf(__hip_bfloat16* p, long i, float v) { p[i] = __float2bfloat16(v); }
constexpr double NEG_INF = -__builtin_huge_val();

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// ------------------------------------------------------------------------------------------------
// per-cell log-sum-exp, separable logits.  grid (ceil(T/32), ceil(U1/8), B), 256 threads = 32 t x 8 u
// ------------------------------------------------------------------------------------------------
constexpr int LSE_UT = 8, LSE_VC = 128;

__global__ void __launch_bounds__(256) lse_sep_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                      const float* __restrict__ bias, const int* __restrict__ labels,
                                                      int T, int U1, int V, int blank, long a_sb, long a_st,
                                                      long c_sb, long c_su, float* __restrict__ blk,
                                                      float* __restrict__ emit) {
  __shared__ float As[TT][LSE_VC + 1];
  __shared__ float Cs[LSE_UT][LSE_VC + 1];
  const int b = blockIdx.z, t0 = blockIdx.x * TT, u0 = blockIdx.y * LSE_UT;
  const int tid = threadIdx.x, tl = tid & 31, ul = tid >> 5;
  const int t = t0 + tl, u = u0 + ul;
  const float* Ab = A + (long)b * a_sb;
  const float* Cb = C + (long)b * c_sb;

  float m = -__builtin_huge_valf(), s = 0.f;
  for (int v0 = 0; v0 < V; v0 += LSE_VC) {
    const int vc = min(LSE_VC, V - v0);
    __syncthreads();
    for (int i = tid; i < TT * LSE_VC; i += 256) {
      const int r = i / LSE_VC, c = i % LSE_VC;
      As[r][c] = (t0 + r < T && c < vc) ? Ab[(long)(t0 + r) * a_st + v0 + c] : 0.f;
    }
    for (int i = tid; i < LSE_UT * LSE_VC; i += 256) {
      const int r = i / LSE_VC, c = i % LSE_VC;
      Cs[r][c] = (u0 + r < U1 && c < vc) ? Cb[(long)(u0 + r) * c_su + v0 + c] + bias[v0 + c] : 0.f;
    }
    __syncthreads();
    float cm = -__builtin_huge_valf();
    for (int c = 0; c < vc; ++c) cm = fmaxf(cm, As[tl][c] + Cs[ul][c]);
    const float nm = fmaxf(m, cm);
    float cs = 0.f;
    for (int c = 0; c < vc; ++c) cs += expf(As[tl][c] + Cs[ul][c] - nm);
    s = s * expf(m - nm) + cs;
    m = nm;
  }
  if (t < T && u < U1) {
    const float lse = m + logf(s);
    const long o = ((long)b * U1 + u) * T + t;
    const float zb = Ab[(long)t * a_st + blank] + Cb[(long)u * c_su + blank] + bias[blank];
    blk[o] = zb - lse;
    float e = 0.f;
    if (u < U1 - 1) {
      const int y = labels[(long)b * (U1 - 1) + u];
      e = Ab[(long)t * a_st + y] + Cb[(long)u * c_su + y] + bias[y] - lse;
    }
    emit[o] = e;
  }
}


// ------------------------------------------------------------------------------------------------
// per-cell log-sum-exp for LARGE vocabularies (V >= 256: BASELINE configs[4], V = 2048).  Same outputs as lse_sep_kernel.
// lse_sep_kernel spends its time in ocml expf (~20 VALU instructions per element) and in two LDS reads per element; here a thread
// owns 4 label positions of one frame (every A value read from LDS feeds 4 cells), the chunk is staged pre-multiplied by log2(e) and
// the sum runs on v_exp_f32 (2^x): per element one add for the chunk maximum, then add + sub + v_exp_f32 + add.
// grid (ceil(T/32), ceil(U1/32), B), 256 threads = 32 t x 8 groups of 4 u.
// ------------------------------------------------------------------------------------------------
constexpr int LSV_UR = 4, LSV_UT = 8 * LSV_UR, LSV_VC = 128;
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;

__global__ void __launch_bounds__(256) lse_sepv_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                       const float* __restrict__ bias, const int* __restrict__ labels,
                                                       int T, int U1, int V, int blank, long a_sb, long a_st,
                                                       long c_sb, long c_su, float* __restrict__ blk,
                                                       float* __restrict__ emit) {
  __shared__ float As[TT][LSV_VC + 1];
  __shared__ float Cs[LSV_UT][LSV_VC + 1];
  const int b = blockIdx.z, t0 = blockIdx.x * TT, u0 = blockIdx.y * LSV_UT;
  const int tid = threadIdx.x, tl = tid & 31, ug = tid >> 5;
  const int t = t0 + tl;
  const float* Ab = A + (long)b * a_sb;
  const float* Cb = C + (long)b * c_sb;

  float m[LSV_UR], s[LSV_UR];   // running maximum (log2 domain) and sum of 2^(x - m)
#pragma unroll
  for (int r = 0; r < LSV_UR; ++r) { m[r] = -__builtin_huge_valf(); s[r] = 0.f; }
  for (int v0 = 0; v0 < V; v0 += LSV_VC) {
    const int vc = min(LSV_VC, V - v0);
    __syncthreads();
    for (int i = tid; i < TT * LSV_VC; i += 256) {
      const int r = i / LSV_VC, c = i % LSV_VC;
      As[r][c] = (t0 + r < T && c < vc) ? Ab[(long)(t0 + r) * a_st + v0 + c] * LOG2E_F : 0.f;
    }
    for (int i = tid; i < LSV_UT * LSV_VC; i += 256) {
      const int r = i / LSV_VC, c = i % LSV_VC;
      Cs[r][c] = (u0 + r < U1 && c < vc) ? (Cb[(long)(u0 + r) * c_su + v0 + c] + bias[v0 + c]) * LOG2E_F : 0.f;
    }
    __syncthreads();
    float cm[LSV_UR];
#pragma unroll
    for (int r = 0; r < LSV_UR; ++r) cm[r] = -__builtin_huge_valf();
    for (int c = 0; c < vc; ++c) {
      const float a = As[tl][c];
#pragma unroll
      for (int r = 0; r < LSV_UR; ++r) cm[r] = fmaxf(cm[r], a + Cs[ug * LSV_UR + r][c]);
    }
    float nm[LSV_UR], cs[LSV_UR];
#pragma unroll
    for (int r = 0; r < LSV_UR; ++r) { nm[r] = fmaxf(m[r], cm[r]); cs[r] = 0.f; }
    for (int c = 0; c < vc; ++c) {
      const float a = As[tl][c];
#pragma unroll
      for (int r = 0; r < LSV_UR; ++r) cs[r] += __builtin_amdgcn_exp2f(a + Cs[ug * LSV_UR + r][c] - nm[r]);
    }
#pragma unroll
    for (int r = 0; r < LSV_UR; ++r) {
      s[r] = s[r] * __builtin_amdgcn_exp2f(m[r] - nm[r]) + cs[r];   // (first chunk: s = 0, 2^-inf = 0)
      m[r] = nm[r];
    }
  }
#pragma unroll
  for (int r = 0; r < LSV_UR; ++r) {
    const int u = u0 + ug * LSV_UR + r;
    if (t < T && u < U1) {
      const float lse = (m[r] + __builtin_amdgcn_logf(s[r])) * LN2_F;   // v_log_f32 = log2
      const long o = ((long)b * U1 + u) * T + t;
      const float zb = Ab[(long)t * a_st + blank] + Cb[(long)u * c_su + blank] + bias[blank];
      blk[o] = zb - lse;
      float e = 0.f;
      if (u < U1 - 1) {
        const int y = labels[(long)b * (U1 - 1) + u];
        e = Ab[(long)t * a_st + y] + Cb[(long)u * c_su + y] + bias[y] - lse;
      }
      emit[o] = e;
    }
  }
}

// dense logits (B,T,U1,V): one wavefront per cell, lanes along v, grid-stride over cells
template <typename TZ>
__global__ void __launch_bounds__(256) lse_dense_kernel(const TZ* __restrict__ Z, const int* __restrict__ labels,
                                                        int B, int T, int U1, int V, int blank,
                                                        float* __restrict__ blk, float* __restrict__ emit) {
  const int lane = threadIdx.x & 63;
  const long ncell = (long)B * T * U1;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long cell = wave0; cell < ncell; cell += nw) {
    const TZ* z = Z + cell * V;
    float m = -__builtin_huge_valf();
    for (int v = lane; v < V; v += 64) m = fmaxf(m, ldf(z, v));
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(ldf(z, v) - m);
    s = wave_sum(s);
    if (lane == 0) {
      const int u = (int)(cell % U1);
      const long bt = cell / U1;
      const int t = (int)(bt % T), b = (int)(bt / T);
      const float lse = m + logf(s);
      const long o = ((long)b * U1 + u) * T + t;
      blk[o] = ldf(z, blank) - lse;
      emit[o] = (u < U1 - 1) ? ldf(z, labels[(long)b * (U1 - 1) + u]) - lse : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// alpha / beta: one wavefront per (b, which).  fp64 accumulation, fp32 correction term.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double logaddexp_d(double a, double b) {
  const double m = fmax(a, b);
  if (m == NEG_INF) return NEG_INF;
  const float d = (float)(fmin(a, b) - m);  // <= 0, -inf allowed
  // fp32 correction term in [0, ln 2]: hardware exp/log (abs err ~1e-7) -- the fp64 running sums keep the lattice exact
  return m + (double)__logf(1.0f + __expf(d));
}
// neighbour hand-off across the whole wavefront by DPP (wave_shr:1 / wave_shl:1): two moves per fp64, no LDS permute
__device__ __forceinline__ double shfl_up1(double x, int lane) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, false);
  const double y = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  return lane == 0 ? NEG_INF : y;
}
__device__ __forceinline__ double shfl_down1(double x, int lane) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, false);
  const double y = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  return lane == 63 ? NEG_INF : y;
}

constexpr int PF = 8;  // rows of blk/emit in flight per lane (register ring)
constexpr int AB_RSRC = 0x00027000, AB_OOB = 0x7ffffff0;
// The sweep is a chain of ~T + U dependent steps (one anti-diagonal each).  Every memory operation of the loop goes through a buffer
// resource with an out-of-range offset for lanes / steps that have no cell (loads return 0, stores are dropped): the loop body has NO
// branch around a memory operation, so hipcc counts how many younger operations may stay in flight when it waits for a ring slot
// (`s_waitcnt vmcnt(n)`, n > 0).  With the loads and stores inside `if (cell exists)` branches — the round-1 form — it could not, fell
// back to vmcnt(0) after every step and each step waited out the loads it had just issued for 8 steps later: 950 cycles per step at
// config 2 (0.41 ms per training step for a 1 040-step chain) against the ~300 the arithmetic takes.
template <int K>
__global__ void __launch_bounds__(64) alphabeta_kernel(const float* __restrict__ blk, const float* __restrict__ emit,
                                                       const int* __restrict__ t_lens, const int* __restrict__ u_lens,
                                                       int T, int U1, double* __restrict__ alpha,
                                                       double* __restrict__ beta, double* __restrict__ ll) {
  const int b = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
  const int Tb = t_lens[b], Ub = u_lens[b];  // valid cells: t < Tb, u <= Ub
  const long rowbase = (long)b * U1 * T;
  const int ulo = lane * K;
  const int lastlane = Ub / K;
  const int nsteps = Tb + lastlane;
  const int cells = U1 * T;
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blk + rowbase), 0, cells * 4, AB_RSRC);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(emit + rowbase), 0, cells * 4, AB_RSRC);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((which == 0 ? alpha : beta) + rowbase, 0, cells * 8, AB_RSRC);
  auto cell = [&](int u, int t) -> int { return (t >= 0 && t < Tb && u <= Ub) ? u * T + t : -1; };
  auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int c) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, c >= 0 ? c * 4 : AB_OOB, 0, 0));
  };
  auto st = [&](double v, int c) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, v), ro, c >= 0 ? c * 8 : AB_OOB, 0, 0);
  };
  const int nrounds = (nsteps + PF - 1) / PF;   // steps beyond nsteps touch no cell (every lane is past its last row)
  double llv = NEG_INF;

  if (which == 0) {
    double down[K];  // alpha[t-1][u] + blk[t-1][u]
#pragma unroll
    for (int k = 0; k < K; ++k) down[k] = NEG_INF;
    double eout = NEG_INF;  // alpha[t][uhi] + emit[t][uhi] of this lane's last cell at its current row
    // a row's blk/emit values are fetched PF steps ahead into a register ring
    float pb[PF][K], pe[PF][K];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c = cell(ulo + k, j - lane);
        pb[j][k] = ld(rb, c);
        pe[j][k] = ld(re, c);
      }
    for (int r = 0; r < nrounds; ++r) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int t = r * PF + j - lane;
        const double carry = shfl_up1(eout, lane);
        float cb[K], ce[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { cb[k] = pb[j][k]; ce[k] = pe[j][k]; }
        // refill this ring slot with the row of step n + PF
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int c = cell(ulo + k, t + PF);
          pb[j][k] = ld(rb, c);
          pe[j][k] = ld(re, c);
        }
        const bool row = t >= 0 && t < Tb;
        double left = carry;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int u = ulo + k;
          const bool ok = row && u <= Ub;
          const double a = (t == 0 && u == 0) ? 0.0 : logaddexp_d(down[k], left);
          st(a, ok ? u * T + t : -1);
          const double nd = a + (double)cb[k];
          down[k] = ok ? nd : down[k];
          left = ok ? ((u < Ub) ? a + (double)ce[k] : NEG_INF) : left;
          llv = (ok && t == Tb - 1 && u == Ub) ? nd : llv;
        }
        eout = row ? left : eout;
      }
    }
    if (lane == lastlane) ll[b] = llv;   // the lane that owns cell (Tb - 1, Ub)
  } else {
    double down[K];  // beta[t+1][u]
#pragma unroll
    for (int k = 0; k < K; ++k) down[k] = NEG_INF;
    double bout = NEG_INF;  // beta[t][ulo] of this lane's first cell at its current row
    // lane l at step n handles t = Tb-1 - (n - (lastlane - l)); lanes > lastlane own no valid cell
    float pb[PF][K], pe[PF][K];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int c = cell(ulo + k, Tb - 1 - (j - (lastlane - lane)));
        pb[j][k] = ld(rb, c);
        pe[j][k] = ld(re, c);
      }
    for (int r = 0; r < nrounds; ++r) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int t = Tb - 1 - (r * PF + j - (lastlane - lane));
        const double carry = shfl_down1(bout, lane);  // beta[t][(lane+1)*K]
        float cb[K], ce[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { cb[k] = pb[j][k]; ce[k] = pe[j][k]; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int c = cell(ulo + k, t - PF);
          pb[j][k] = ld(rb, c);
          pe[j][k] = ld(re, c);
        }
        const bool row = t >= 0 && t < Tb && lane <= lastlane;
        double right = carry;  // beta[t][u+1]
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
          const int u = ulo + k;
          const bool ok = row && u <= Ub;
          const double ne = down[k] == NEG_INF ? NEG_INF : down[k] + (double)cb[k];
          const double em = (u < Ub && right != NEG_INF) ? right + (double)ce[k] : NEG_INF;
          const double v = (t == Tb - 1 && u == Ub) ? (double)cb[k] : logaddexp_d(ne, em);
          st(v, ok ? u * T + t : -1);
          down[k] = ok ? v : down[k];
          right = ok ? v : right;
          llv = (ok && t == 0 && u == 0) ? v : llv;
        }
        bout = row ? right : bout;
      }
    }
    if (lane == 0) ll[gridDim.x + b] = llv;   // the lane that owns cell (0, 0)
  }
}

// ------------------------------------------------------------------------------------------------
// per-cell scalars for the gradient: occupancy w, row lse, blank- and label-transition posteriors
// ------------------------------------------------------------------------------------------------
struct CellS {
  float w, lse, cb, ce;
};

__device__ __forceinline__ CellS cell_scalars(const float* __restrict__ blk, const float* __restrict__ emit,
                                              const double* __restrict__ alpha, const double* __restrict__ beta,
                                              long rowbase, int T, int t, int u, int Tb, int Ub, double logZ,
                                              float zblank) {
  // (t, u) must be a cell of the lattice (t < Tb, u <= Ub).  All six loads are unconditional — the neighbours' indices are clamped
  // to the cell itself where the lattice ends and the value is dropped by a select — so a caller that builds several cells per thread
  // gets its loads issued back to back instead of one dependent round trip per `if`.
  CellS c;
  const long o = rowbase + (long)u * T + t;
  const bool last_t = t == Tb - 1, has_label = u < Ub;
  const double a = alpha[o], bt = beta[o];
  const double b_next_t = beta[last_t ? o : o + 1], b_next_u = beta[has_label ? o + T : o];
  const float lb = blk[o], le = emit[o];
  c.w = expf((float)(a + bt - logZ));
  c.lse = zblank - lb;
  const double tb = last_t ? ((u == Ub) ? a + (double)lb - logZ : NEG_INF) : a + (double)lb + b_next_t - logZ;
  c.cb = expf((float)tb);
  c.ce = has_label ? expf((float)(a + (double)le + b_next_u - logZ)) : 0.f;
  return c;
}

// grid (ceil(T/32), B), 256 threads; wave w owns frames w, w+4, ... of the tile; lanes run along v.
// dynamic LDS: cells[TT][U1] (CellS) | Cs[U1][64] | dCs[U1][64] | ys[U1]
__global__ void __launch_bounds__(256) grad_sep_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                       const float* __restrict__ bias, const int* __restrict__ labels,
                                                       const int* __restrict__ t_lens, const int* __restrict__ u_lens,
                                                       const float* __restrict__ blk, const float* __restrict__ emit,
                                                       const double* __restrict__ alpha, const double* __restrict__ beta,
                                                       const double* __restrict__ ll, int T, int U1, int V, int blank,
                                                       long a_sb, long a_st, long c_sb, long c_su, float gscale_in,
                                                       const float* __restrict__ gvec, int gvec_stride,
                                                       float* __restrict__ dA, float* __restrict__ dCp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CellS* cells = reinterpret_cast<CellS*>(smem);
  float* Cs = reinterpret_cast<float*>(cells + TT * U1);
  float* dCs = Cs + U1 * 64;
  int* ys = reinterpret_cast<int*>(dCs + U1 * 64);

  const int b = blockIdx.y, tile = blockIdx.x, t0 = tile * TT, ntiles = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = t_lens[b], Ub = u_lens[b];
  const long rowbase = (long)b * U1 * T;
  const double logZ = ll[b];
  const float gscale = gvec ? gscale_in * gvec[(long)b * gvec_stride] : gscale_in;  // upstream gradient per utterance (stride 0: one scalar)
  const float* Ab = A + (long)b * a_sb;
  const float* Cb = C + (long)b * c_sb;
  float* dCtile = dCp + ((long)b * ntiles + tile) * U1 * V;

  for (int i = tid; i < U1; i += 256) ys[i] = (i < U1 - 1) ? labels[(long)b * (U1 - 1) + i] : -1;
  for (int i = tid; i < TT * U1; i += 256) {
    const int u = i / TT, tl = i % TT, t = t0 + tl;   // consecutive threads -> consecutive frames: alpha / beta / blk / emit rows are read coalesced
    CellS c = {0.f, 0.f, 0.f, 0.f};
    if (t < Tb && u <= Ub) {
      const float zb = Ab[(long)t * a_st + blank] + Cb[(long)u * c_su + blank] + bias[blank];
      c = cell_scalars(blk, emit, alpha, beta, rowbase, T, t, u, Tb, Ub, logZ, zb);
    }
    cells[tl * U1 + u] = c;
  }

  for (int v0 = 0; v0 < V; v0 += 64) {
    const int v = v0 + lane;
    const bool vok = v < V;
    __syncthreads();  // cells/ys ready (first pass); previous chunk's dCs flushed (later passes)
    for (int i = tid; i < U1 * 64; i += 256) {
      const int u = i >> 6, c = i & 63;
      Cs[i] = (v0 + c < V) ? Cb[(long)u * c_su + v0 + c] + bias[v0 + c] : 0.f;
      dCs[i] = 0.f;
    }
    __syncthreads();
    // A wave owns the frames tl = wave + 4*i (i < TT/4) of the tile and walks u in chunks of UC: dA of a frame (sum over
    // u) and dC of a label position (sum over the wave's frames) both accumulate in REGISTERS; the LDS adds that merge the
    // four waves' dC happen once per (u, wave) instead of once per (u, frame) -- 8x fewer, and they were the kernel's
    // bottleneck at large V (c5: V = 2048) -- and in a fixed order.
    constexpr int NF = TT / 4, UC = 8;
    float a[NF], accA[NF];
    bool fok[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int t = t0 + wave + 4 * i;
      fok[i] = t < Tb && vok;
      a[i] = fok[i] ? Ab[(long)t * a_st + v] : 0.f;
      accA[i] = 0.f;
    }
    for (int u0 = 0; u0 <= Ub; u0 += UC) {
      float accC[UC], cs[UC];
      int yv[UC];
#pragma unroll
      for (int j = 0; j < UC; ++j) {
        const int u = min(u0 + j, U1 - 1);
        accC[j] = 0.f;
        cs[j] = Cs[u * 64 + lane];
        yv[j] = ys[u];
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        if (!fok[i]) continue;
        const CellS* crow = cells + (wave + 4 * i) * U1;
#pragma unroll
        for (int j = 0; j < UC; ++j) {
          const int u = u0 + j;
          if (u > Ub) break;
          const CellS c = crow[u];
          // v_exp_f32 form: |rel err| <~ 3e-7 on a value in [0, 1] (the softmax probability times a path weight <= 1), i.e. an
          // absolute gradient error ~1e-7 against the 2e-5 the tests allow; ocml expf was 2/3 of this loop's instructions
          float g = c.w * __expf(a[i] + cs[j] - c.lse);
          if (v == blank) g -= c.cb;
          if (v == yv[j]) g -= c.ce;
          accA[i] += g;
          accC[j] += g;
        }
      }
      // merge the four waves' sums in wave order: fixed order -> the step is bitwise reproducible (LDS float atomics were not);
      // u0 / Ub are uniform over the workgroup, so every wave meets every barrier
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (wave == w && vok) {
#pragma unroll
          for (int j = 0; j < UC; ++j)
            if (u0 + j <= Ub) dCs[(u0 + j) * 64 + lane] += accC[j];
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int t = t0 + wave + 4 * i;
      if (t < T && vok) dA[(long)b * a_sb + (long)t * a_st + v] = accA[i] * gscale;
    }
    __syncthreads();
    for (int i = tid; i < U1 * 64; i += 256) {
      const int u = i >> 6, c = i & 63;
      if (v0 + c < V) dCtile[(long)u * V + v0 + c] = dCs[i];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// lattice gradient for LARGE vocabularies (V >= 256).  Same outputs as grad_sep_kernel (dA, per-tile dC slabs for reduce_dc_kernel).
// grad_sep_kernel keeps the whole vocabulary loop inside one workgroup per (utterance, 32-frame tile): 82 KB of LDS (one workgroup of
// 4 waves per CU), four barriers and an LDS merge of the waves' dC sums per 8 label positions and 64-entry vocabulary chunk — at
// V = 2048 it ran 5.6 ms for 4 G elements (profiles/r02_final_bench_c5.log), 7x off the v_exp_f32 rate.  Here the vocabulary is a
// GRID dimension and a LANE owns one vocabulary entry v for the whole tile: dA[t][v] (sum over u) and dC[u][v] (sum over the tile's
// frames) are both plain register sums of that lane — no cross-lane or cross-wave merge, no barrier after the cell table is built,
// LDS holds the per-cell scalars only (read as broadcasts).  The blank / label corrections leave the inner loop: the one wave whose
// 64 entries contain the blank (a label) subtracts them in a second, exp-free pass.  Arithmetic per element: two adds, v_exp_f32
// (2^x on operands pre-multiplied by log2 e), one multiply, two accumulations.
// grid (ceil(T/32), ceil(V / (64 NW)), B), 64 NW threads; dynamic LDS: cells[TT][U1] (CellS, lse in the log2 domain) | ys[U1]
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(64 * NW, 3) grad_sepv_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                            const float* __restrict__ bias, const int* __restrict__ labels,
                                                            const int* __restrict__ t_lens, const int* __restrict__ u_lens,
                                                            const float* __restrict__ blk, const float* __restrict__ emit,
                                                            const double* __restrict__ alpha, const double* __restrict__ beta,
                                                            const double* __restrict__ ll, int T, int U1, int V, int blank,
                                                            long a_sb, long a_st, long c_sb, long c_su, float gscale_in,
                                                            const float* __restrict__ gvec, int gvec_stride,
                                                            float* __restrict__ dA, float* __restrict__ dCp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* wl = reinterpret_cast<float2*>(smem);          // [TT][U1] {occupancy w, row lse * log2 e}: what the inner loop reads (8-byte broadcasts)
  float2* cbe = wl + TT * U1;                            // [TT][U1] {P(blank transition), P(label transition)}: the correction pass
  int* ys = reinterpret_cast<int*>(cbe + TT * U1);

  const int b = blockIdx.z, tile = blockIdx.x, t0 = tile * TT, ntiles = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vw0 = (blockIdx.y * NW + wave) * 64;     // first vocabulary entry of this wave
  const int v = vw0 + lane;
  const bool vok = v < V;
  const int Tb = t_lens[b], Ub = u_lens[b];
  const long rowbase = (long)b * U1 * T;
  const double logZ = ll[b];
  const float gscale = gvec ? gscale_in * gvec[(long)b * gvec_stride] : gscale_in;
  const float* Ab = A + (long)b * a_sb;
  const float* Cb = C + (long)b * c_sb;
  float* dCtile = dCp + ((long)b * ntiles + tile) * U1 * V;
  const int nf = max(0, min(TT, Tb - t0));            // valid frames of this tile (uniform)

  for (int i = tid; i < U1; i += 64 * NW) ys[i] = (i < U1 - 1) ? labels[(long)b * (U1 - 1) + i] : -1;
  // the cell table: four cells per thread and round, every load unconditional (indices clamped into the lattice, results dropped by
  // selects) so that the 4 x 8 loads of a round are in flight together; consecutive threads -> consecutive frames (coalesced rows)
  const float bias_blank = bias[blank];
  for (int i0 = tid; i0 < TT * U1; i0 += 64 * NW * 4) {
    CellS c[4];
    bool in[4];
    int dst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * 64 * NW;
      const int u = min(i / TT, U1 - 1), tl = i % TT, t = t0 + tl;
      in[q] = i < TT * U1 && t < Tb && u <= Ub;
      dst[q] = i < TT * U1 ? tl * U1 + u : -1;
      const int tc = min(t, Tb - 1), uc = min(u, Ub);   // a cell of the lattice whatever (t, u) is
      const float zb = Ab[(long)tc * a_st + blank] + Cb[(long)uc * c_su + blank] + bias_blank;
      c[q] = cell_scalars(blk, emit, alpha, beta, rowbase, T, tc, uc, Tb, Ub, logZ, zb);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (dst[q] >= 0) {
        // cells outside the utterance's lattice: w = 0 and an lse that sends 2^(x - lse) to 0, so the inner loop needs no per-cell test
        wl[dst[q]] = in[q] ? make_float2(c[q].w, c[q].lse * LOG2E_F) : make_float2(0.f, 1e30f);
        cbe[dst[q]] = in[q] ? make_float2(c[q].cb, c[q].ce) : make_float2(0.f, 0.f);
      }
    }
  }
  __syncthreads();
  if (vw0 >= V) return;   // (whole wave beyond the vocabulary: nothing to do, and no barrier follows)

  float a2[TT], accA[TT];
#pragma unroll
  for (int tl = 0; tl < TT; ++tl) {
    a2[tl] = (tl < nf && vok) ? Ab[(long)(t0 + tl) * a_st + v] * LOG2E_F : 0.f;
    accA[tl] = 0.f;
  }
  const float bias_v = vok ? bias[v] : 0.f;
  const bool has_blank = blank >= vw0 && blank < vw0 + 64;   // uniform over the wave
  constexpr int UC = 8;
  for (int u0 = 0; u0 < U1; u0 += UC) {
    float accC[UC], cs2[UC];
#pragma unroll
    for (int j = 0; j < UC; ++j) {
      const int u = min(u0 + j, U1 - 1);
      accC[j] = 0.f;
      cs2[j] = vok ? (Cb[(long)u * c_su + v] + bias_v) * LOG2E_F : 0.f;
    }
    if (u0 <= Ub) {
      // branch-free over the whole 32 x 8 block (cells outside the lattice contribute exact zeros; label positions beyond U1 - 1
      // re-read the last row's table entries, whose sums are not stored): the 256 table reads and v_exp_f32 pipeline freely
#pragma unroll
      for (int tl = 0; tl < TT; ++tl) {
#pragma unroll
        for (int j = 0; j < UC; ++j) {
          const float2 c = wl[tl * U1 + min(u0 + j, U1 - 1)];
          // the softmax probability (<= 1: the exponent is z - lse <= 0) times the cell's occupancy
          const float g = c.x * __builtin_amdgcn_exp2f(a2[tl] + cs2[j] - c.y);
          accA[tl] += (u0 + j < U1) ? g : 0.f;   // (a -1e30 entry in cs2 instead of this select: hipcc then spilled — 3.3 instead of 2.1 ms)
          accC[j] += g;
        }
        // two frames (16 table reads) per scheduling region: without the fence hipcc hoists all 256 reads of the block to its top
        // (288 registers: one wave per SIMD; with a register cap: spills), with it 131 registers and three waves per SIMD cover the reads
        if (tl & 1) __builtin_amdgcn_sched_barrier(0);
      }
      // corrections: - P(blank transition) on the blank entry, - P(label transition) on the label's entry
#pragma unroll
      for (int j = 0; j < UC; ++j) {
        const int u = u0 + j;
        if (u <= Ub) {
          const int y = ys[u];
          const bool mine = y >= vw0 && y < vw0 + 64;   // uniform
          if (mine || has_blank) {
#pragma unroll
            for (int tl = 0; tl < TT; ++tl) {
              if (tl < nf) {
                const float2 c = cbe[tl * U1 + u];
                const float corr = (v == blank ? c.x : 0.f) + (v == y ? c.y : 0.f);
                accA[tl] -= corr;
                accC[j] -= corr;
              }
            }
          }
        }
      }
    }
    if (vok) {
#pragma unroll
      for (int j = 0; j < UC; ++j)
        if (u0 + j < U1) dCtile[(long)(u0 + j) * V + v] = accC[j];   // rows beyond Ub: zeros (reduce_dc_kernel sums every row)
    }
  }
  if (vok) {
#pragma unroll
    for (int tl = 0; tl < TT; ++tl)
      if (t0 + tl < T) dA[(long)b * a_sb + (long)(t0 + tl) * a_st + v] = accA[tl] * gscale;
  }
}

__global__ void __launch_bounds__(256) reduce_dc_kernel(const float* __restrict__ dCp, int ntiles, long per_b, int V,
                                                        long c_sb, long c_su, float gscale_in, const float* __restrict__ gvec,
                                                        int gvec_stride, float* __restrict__ dC) {
  const int b = blockIdx.y;
  const float gscale = gvec ? gscale_in * gvec[(long)b * gvec_stride] : gscale_in;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_b) return;
  const float* src = dCp + (long)b * ntiles * per_b + i;
  float s = 0.f;
  for (int k = 0; k < ntiles; ++k) s += src[(long)k * per_b];
  dC[(long)b * c_sb + (i / V) * c_su + (i % V)] = s * gscale;
}

// dense d/d logits: one wavefront per cell
template <typename TZ>
__global__ void __launch_bounds__(256) grad_dense_kernel(const TZ* __restrict__ Z, const int* __restrict__ labels,
                                                         const int* __restrict__ t_lens, const int* __restrict__ u_lens,
                                                         const float* __restrict__ blk, const float* __restrict__ emit,
                                                         const double* __restrict__ alpha, const double* __restrict__ beta,
                                                         const double* __restrict__ ll, int B, int T, int U1, int V,
                                                         int blank, float gscale, TZ* __restrict__ G) {
  const int lane = threadIdx.x & 63;
  const long ncell = (long)B * T * U1;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long cell = wave0; cell < ncell; cell += nw) {
    const int u = (int)(cell % U1);
    const long bt = cell / U1;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int Tb = t_lens[b], Ub = u_lens[b];
    const TZ* z = Z + cell * V;
    TZ* g = G + cell * V;
    if (t >= Tb || u > Ub) {
      for (int v = lane; v < V; v += 64) stf(g, v, 0.f);
      continue;
    }
    const CellS c = cell_scalars(blk, emit, alpha, beta, (long)b * U1 * T, T, t, u, Tb, Ub, ll[b], ldf(z, blank));
    const int y = (u < U1 - 1) ? labels[(long)b * (U1 - 1) + u] : -1;
    for (int v = lane; v < V; v += 64) {
      float x = c.w * expf(ldf(z, v) - c.lse);
      if (v == blank) x -= c.cb;
      if (v == y && u < Ub) x -= c.ce;
      stf(g, v, x * gscale);
    }
  }
}

__global__ void __launch_bounds__(256) joint_logits_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                           const float* __restrict__ bias, int T, int U1, int V,
                                                           long a_sb, long a_st, long c_sb, long c_su, long total,
                                                           float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % V);
    long r = i / V;
    const int u = (int)(r % U1);
    r /= U1;  // r = b*T + t
    const long b = r / T, t = r % T;
    out[i] = A[b * a_sb + t * a_st + v] + C[b * c_sb + u * c_su + v] + bias[v];
  }
}

__global__ void nll_kernel(const double* __restrict__ ll, int B, float* __restrict__ nll) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) nll[b] = (float)(-ll[b]);
}

// out[0] = scale * sum_i x[i]: one wavefront, fixed order (lane partial sums over i = lane, lane + 64, ... then a butterfly)
__global__ void __launch_bounds__(64) scaled_sum_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += x[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}

struct LossWs {
  float *blk, *emit;
  double *alpha, *beta, *ll;
  float* dCp;
  size_t total;
};

LossWs carve(void* ws, int B, int T, int U1, int V, bool with_slabs) {
  LossWs w;
  const size_t cells = (size_t)B * T * U1;
  char* p = reinterpret_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += align_up(bytes, 256); return q; };
  w.alpha = reinterpret_cast<double*>(take(cells * 8));
  w.beta = reinterpret_cast<double*>(take(cells * 8));
  w.ll = reinterpret_cast<double*>(take((size_t)B * 2 * 8));
  w.blk = reinterpret_cast<float*>(take(cells * 4));
  w.emit = reinterpret_cast<float*>(take(cells * 4));
  const size_t ntiles = ceil_div(T, TT);
  w.dCp = reinterpret_cast<float*>(take(with_slabs ? (size_t)B * ntiles * U1 * V * 4 : 0));
  w.total = off;
  return w;
}

int check_common(const void* labels, const void* t_lens, const void* u_lens, int B, int T, int U1, int V, int blank,
                 const void* nll) {
  RNNT_CHECK_ARG(B >= 1 && T >= 1 && U1 >= 1 && V >= 1, "rnnt loss: dims must be positive (B=%d T=%d U1=%d V=%d)", B, T, U1, V);
  RNNT_CHECK_ARG(blank >= 0 && blank < V, "rnnt loss: blank %d outside [0,%d)", blank, V);
  RNNT_CHECK_ARG(U1 <= 64 * 8, "rnnt loss: U+1 = %d exceeds the 512 label positions one wavefront sweeps", U1);
  RNNT_CHECK_ARG((int64_t)U1 * T * 8 < (1ll << 31), "rnnt loss: one utterance's lattice (T = %d x U+1 = %d, fp64) exceeds the 2 GB a buffer resource addresses", T, U1);
  RNNT_CHECK_ARG(t_lens && u_lens && nll, "rnnt loss: null lengths/output");
  RNNT_CHECK_ARG(U1 == 1 || labels, "rnnt loss: null labels");
  return RNNT_OK;
}

int launch_alphabeta(const LossWs& w, const int* t_lens, const int* u_lens, int B, int T, int U1, hipStream_t s) {
  const int K = (int)ceil_div(U1, 64);
  ProfScope prof(RNNT_K_ALPHABETA, 2.0 * (8.0 + 8.0) * (double)B * T * U1, s);  // read blk+emit, write alpha|beta (fp64)
  dim3 grid(B, 2), block(64);
#define AB(KK) hipLaunchKernelGGL((alphabeta_kernel<KK>), grid, block, 0, s, w.blk, w.emit, t_lens, u_lens, T, U1, w.alpha, w.beta, w.ll)
  switch (K) {
    case 1: AB(1); break;
    case 2: AB(2); break;
    case 3: AB(3); break;
    case 4: AB(4); break;
    default: AB(8); break;
  }
#undef AB
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

}  // namespace
}  // namespace rnnt

using namespace rnnt;

// vocabularies from 256 entries take the lane-per-entry kernels (lse_sepv_kernel, grad_sepv_kernel); RNNT_LOSS_SMALLV_KERNELS=1 keeps
// the round-1 kernels everywhere, RNNT_LOSS_LARGEV_KERNELS=1 takes the new ones for every V (A/B and tests)
static bool large_vocab(int V) {
  if (getenv("RNNT_LOSS_SMALLV_KERNELS")) return false;
  return V >= 256 || getenv("RNNT_LOSS_LARGEV_KERNELS") != nullptr;
}

static int launch_grad_sep(const LossWs& w, const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb, int64_t c_su,
                           const float* bias, const int32_t* labels, const int32_t* t_lens, const int32_t* u_lens, int32_t B,
                           int32_t T, int32_t U1, int32_t V, int32_t blank, float gscale, const float* gvec, int gvec_stride, float* dA,
                           float* dC, hipStream_t s) {
  const int ntiles = (int)ceil_div(T, TT);
  const double cells = (double)B * T * U1;
  ProfScope prof(RNNT_K_LATGRAD, 4.0 * 2.0 * ((double)B * T * V + (double)B * U1 * V) + 24.0 * cells, s);
  if (large_vocab(V)) {   // a lane per vocabulary entry, the vocabulary a grid dimension (grad_sepv_kernel)
    const size_t lds = (size_t)TT * U1 * sizeof(CellS) + (size_t)U1 * 4;
    RNNT_CHECK_ARG(lds <= 160 * 1024, "joint_loss: U+1 = %d needs %zu B of LDS (> 160 KiB)", U1, lds);
    constexpr int NW = 4;   // 256 vocabulary entries per workgroup (8 waves per workgroup measured slower: 5.2 vs 4.6 ms at config 5)
    if (lds > 64 * 1024)
      RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)grad_sepv_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(grad_sepv_kernel<NW>, dim3(ntiles, (unsigned)ceil_div(V, 64 * NW), B), dim3(64 * NW), lds, s, A, C, bias, labels, t_lens,
                       u_lens, w.blk, w.emit, w.alpha, w.beta, w.ll, T, U1, V, blank, (long)a_sb, (long)a_st, (long)c_sb, (long)c_su, gscale, gvec,
                       gvec_stride, dA, w.dCp);
  } else {
    const size_t lds = (size_t)TT * U1 * sizeof(CellS) + (size_t)U1 * 64 * 4 * 2 + (size_t)U1 * 4;
    RNNT_CHECK_ARG(lds <= 160 * 1024, "joint_loss: U+1 = %d needs %zu B of LDS (> 160 KiB)", U1, lds);
    if (lds > 64 * 1024)
      RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)grad_sep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(grad_sep_kernel, dim3(ntiles, B), dim3(256), lds, s, A, C, bias, labels, t_lens, u_lens, w.blk,
                       w.emit, w.alpha, w.beta, w.ll, T, U1, V, blank, (long)a_sb, (long)a_st, (long)c_sb, (long)c_su, gscale, gvec, gvec_stride, dA, w.dCp);
  }
  RNNT_CHECK_LAUNCH();
  const long per_b = (long)U1 * V;
  hipLaunchKernelGGL(reduce_dc_kernel, dim3((unsigned)ceil_div(per_b, 256), B), dim3(256), 0, s, w.dCp, ntiles, per_b,
                     V, (long)c_sb, (long)c_su, gscale, gvec, gvec_stride, dC);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

extern "C" size_t rnnt_hip_joint_loss_workspace_bytes(int32_t B, int32_t T, int32_t U1, int32_t V) {
  if (B < 1 || T < 1 || U1 < 1 || V < 1) return 0;
  return carve(nullptr, B, T, U1, V, true).total;
}

extern "C" int rnnt_hip_joint_loss_fwd_bwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb,
                                           int64_t c_su, const float* bias, const int32_t* labels,
                                           const int32_t* t_lens, const int32_t* u_lens, int32_t B, int32_t T,
                                           int32_t U1, int32_t V, int32_t blank, float gscale, float* nll, float* dA,
                                           float* dC, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(labels, t_lens, u_lens, B, T, U1, V, blank, nll)) return rc;
  RNNT_CHECK_ARG(A && C && bias, "joint_loss: null A/C/bias");
  RNNT_CHECK_ARG((dA == nullptr) == (dC == nullptr), "joint_loss: dA and dC must both be given or both be NULL");
  const LossWs w = carve(workspace, B, T, U1, V, true);
  RNNT_CHECK_ARG(workspace && workspace_bytes >= w.total, "joint_loss: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = (int)ceil_div(T, TT);

  const double cells = (double)B * T * U1;
  {
  ProfScope prof(RNNT_K_LSE, 4.0 * ((double)B * T * V + (double)B * U1 * V) + 8.0 * cells, s);
  if (large_vocab(V))
    hipLaunchKernelGGL(lse_sepv_kernel, dim3(ntiles, (unsigned)ceil_div(U1, LSV_UT), B), dim3(256), 0, s, A, C, bias, labels,
                       T, U1, V, blank, (long)a_sb, (long)a_st, (long)c_sb, (long)c_su, w.blk, w.emit);
  else
    hipLaunchKernelGGL(lse_sep_kernel, dim3(ntiles, (unsigned)ceil_div(U1, LSE_UT), B), dim3(256), 0, s, A, C, bias, labels,
                       T, U1, V, blank, (long)a_sb, (long)a_st, (long)c_sb, (long)c_su, w.blk, w.emit);
  }
  RNNT_CHECK_LAUNCH();
  if (int rc = launch_alphabeta(w, t_lens, u_lens, B, T, U1, s)) return rc;
  hipLaunchKernelGGL(nll_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, s, w.ll, B, nll);
  RNNT_CHECK_LAUNCH();
  if (dA) return launch_grad_sep(w, A, a_sb, a_st, C, c_sb, c_su, bias, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, nullptr, 0, dA, dC, s);
  return RNNT_OK;
}

extern "C" int rnnt_hip_joint_loss_bwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb, int64_t c_su,
                                       const float* bias, const int32_t* labels, const int32_t* t_lens, const int32_t* u_lens,
                                       int32_t B, int32_t T, int32_t U1, int32_t V, int32_t blank, float gscale,
                                       const float* gvec, int32_t gvec_stride, float* dA, float* dC, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  // second half of rnnt_hip_joint_loss_fwd_bwd: the workspace still holds blk / emit / alpha / beta / logZ of a forward call
  // (dA = dC = NULL) on the SAME A, C, bias, labels, lengths
  RNNT_CHECK_ARG(labels && t_lens && u_lens && B >= 1 && T >= 1 && U1 >= 1 && V >= 2 && blank >= 0 && blank < V,
                 "joint_loss_bwd: bad dims / null pointer");
  RNNT_CHECK_ARG(A && C && bias && dA && dC, "joint_loss_bwd: null A/C/bias/dA/dC");
  RNNT_CHECK_ARG(gvec_stride == 0 || gvec_stride == 1, "joint_loss_bwd: gvec_stride must be 0 (one scalar) or 1 (per utterance)");
  const LossWs w = carve(workspace, B, T, U1, V, true);
  RNNT_CHECK_ARG(workspace && workspace_bytes >= w.total, "joint_loss_bwd: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  return launch_grad_sep(w, A, a_sb, a_st, C, c_sb, c_su, bias, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, gvec, gvec_stride, dA,
                         dC, (hipStream_t)stream);
}

extern "C" int rnnt_hip_scaled_sum_f32(const float* x, int32_t n, float scale, float* out, void* stream) {
  RNNT_CHECK_ARG(x && out && n >= 0, "scaled_sum: bad arguments");
  hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, n, scale, out);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

extern "C" int rnnt_hip_joint_logits_fwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb,
                                         int64_t c_su, const float* bias, int32_t B, int32_t T, int32_t U1, int32_t V,
                                         float* logits, void* stream) {
  RNNT_CHECK_ARG(A && C && bias && logits, "joint_logits: null pointer");
  RNNT_CHECK_ARG(B >= 1 && T >= 1 && U1 >= 1 && V >= 1, "joint_logits: dims must be positive");
  const long total = (long)B * T * U1 * V;
  const unsigned grid = (unsigned)(ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536);
  ProfScope prof(RNNT_K_MISC, 4.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(joint_logits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, C, bias, T, U1, V, (long)a_sb, (long)a_st, (long)c_sb, (long)c_su, total, logits);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

template <typename TZ>
static int loss_from_logits_impl(const void* logits, const int32_t* labels, const int32_t* t_lens, const int32_t* u_lens,
                                 int32_t B, int32_t T, int32_t U1, int32_t V, int32_t blank, float gscale, float* nll,
                                 void* grad, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(labels, t_lens, u_lens, B, T, U1, V, blank, nll)) return rc;
  RNNT_CHECK_ARG(logits, "loss_from_logits: null logits");
  const LossWs w = carve(workspace, B, T, U1, V, true);
  RNNT_CHECK_ARG(workspace && workspace_bytes >= w.total, "loss_from_logits: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  const long ncell = (long)B * T * U1;
  const unsigned grid = (unsigned)(ceil_div(ncell, 4) < 16384 ? ceil_div(ncell, 4) : 16384);
  {
  ProfScope prof(RNNT_K_LSE, (double)sizeof(TZ) * (double)ncell * V + 8.0 * (double)ncell, s);
  hipLaunchKernelGGL((lse_dense_kernel<TZ>), dim3(grid), dim3(256), 0, s, (const TZ*)logits, labels, B, T, U1, V, blank, w.blk, w.emit);
  }
  RNNT_CHECK_LAUNCH();
  if (int rc = launch_alphabeta(w, t_lens, u_lens, B, T, U1, s)) return rc;
  hipLaunchKernelGGL(nll_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, s, w.ll, B, nll);
  RNNT_CHECK_LAUNCH();
  if (grad) {
    ProfScope prof(RNNT_K_LATGRAD, 2.0 * sizeof(TZ) * (double)ncell * V + 24.0 * (double)ncell, s);
    hipLaunchKernelGGL((grad_dense_kernel<TZ>), dim3(grid), dim3(256), 0, s, (const TZ*)logits, labels, t_lens, u_lens, w.blk, w.emit,
                       w.alpha, w.beta, w.ll, B, T, U1, V, blank, gscale, (TZ*)grad);
    RNNT_CHECK_LAUNCH();
  }
  return RNNT_OK;
}

extern "C" int rnnt_hip_loss_from_logits_fwd_bwd_ex(const void* logits, int32_t dtype, const int32_t* labels,
                                                    const int32_t* t_lens, const int32_t* u_lens, int32_t B, int32_t T,
                                                    int32_t U1, int32_t V, int32_t blank, float gscale, float* nll, void* grad,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
  switch (dtype) {
    case RNNT_DTYPE_F32:
      return loss_from_logits_impl<float>(logits, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, nll, grad, workspace, workspace_bytes, stream);
    case RNNT_DTYPE_F16:
      return loss_from_logits_impl<__half>(logits, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, nll, grad, workspace, workspace_bytes, stream);
    case RNNT_DTYPE_BF16:
      return loss_from_logits_impl<__hip_bfloat16>(logits, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, nll, grad, workspace, workspace_bytes, stream);
    default:
      set_error("loss_from_logits: unknown dtype code %d", dtype);
      return RNNT_ERR_INVALID;
  }
}

extern "C" int rnnt_hip_loss_from_logits_fwd_bwd(const float* logits, const int32_t* labels, const int32_t* t_lens,
                                                 const int32_t* u_lens, int32_t B, int32_t T, int32_t U1, int32_t V,
                                                 int32_t blank, float gscale, float* nll, float* grad, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  return rnnt_hip_loss_from_logits_fwd_bwd_ex(logits, RNNT_DTYPE_F32, labels, t_lens, u_lens, B, T, U1, V, blank, gscale, nll, grad,
                                              workspace, workspace_bytes, stream);
}
