// Persistent, length-aware LSTM layer for gfx950: both directions of one layer in ONE launch.
//
// Replaces torch.nn.LSTM over a PackedSequence plus the sort/pack/unpack/unsort around it
// (networks/encoder.py:67-75,93-102; networks/decoder.py:71-79,105-120).
//
// Decomposition:
//   * the input projection X.W_ih^T + b_ih + b_hh for ALL timesteps is hoisted into one MFMA GEMM
//     (gemm.hip) that writes gate pre-activations time-major with the 4 gates of a hidden unit adjacent:
//     gates[t][b][d][4*j + g];
//   * the recurrence runs in one persistent kernel: direction d, hidden slice [j0, j0+Hs) per workgroup
//     (Hs = 4/8/16 units -> D*H/Hs workgroups, one per CU).  The workgroup keeps its 4*Hs rows of W_hh
//     in LDS for the whole sequence, its cell state in registers, and at every step
//       1. waits for the per-workgroup step flags of its direction (sc1 polls, bounded spin),
//       2. gathers h_{t-1} (B x H, write-through exchange buffer, sc1 16-B loads), split-K over the 4 waves,
//          v_mfma_f32_16x16x4_f32 (rows = gate rows, cols = batch),
//       3. reduces the 4 partial tiles through LDS, applies the gate non-linearities, masks t >= len[b],
//       4. writes activated gates (in place of the pre-activations), c_t, y_t and its slice of h_t
//          (sc1 16-B stores), drains, and publishes flag = step+1.
//   * backward mirrors it with dG (B x 4H) as the exchanged vector and v_mfma_f32_4x4x1_16B_f32
//     (16 k-blocks per instruction, so the 4-unit slice wastes no matrix lanes).
//   Packed-sequence semantics need no sort: a row is simply masked while t >= len[b]; its state stays 0,
//   so the reverse direction starts from zero at each sequence's own last frame and padded outputs are 0.
//
// Inter-workgroup protocol: MI355X_MICROARCH "Valid forms" row 1 — every exchanged byte is stored sc1 by
// its owner, every storing wave drains vmcnt(0), workgroup barrier, ONE lane publishes an sc1 flag; the
// consumer polls with sc1 loads from one wave, joins a workgroup barrier, then every wave reads the bytes
// with sc1 loads.  No dispatch-order or XCD-placement assumption; all spins are bounded (status word).
#include "lstm_shared.hpp"

#include <string.h>

namespace rnnt {
namespace {

// ------------------------------------------------------------------------------------------------
// forward recurrence
// dynamic LDS: Wl[4*Hs][LDW] | part[4][MT][NT][64] f32x4 | abort flag
// ------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ void __launch_bounds__(256) lstm_fwd_kernel(const LstmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Wl = reinterpret_cast<float*>(smem);
  f32x4* part = reinterpret_cast<f32x4*>(Wl + 4 * p.Hs * p.LDW);
  int* abort_lds = reinterpret_cast<int*>(part + 4 * MT * NT * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int d = blockIdx.x / p.NC, wg = blockIdx.x % p.NC;
  const int H = p.H, B = p.B, D = p.D, T = p.T, Bp = p.Bp, LDW = p.LDW;
  const int j0 = wg * p.Hs;
  const int R = 4 * p.Hs;

  // stage this workgroup's rows of W_hh: LDS row r = 4*ul + g  <->  torch row g*H + j0 + ul
  {
    const float* W = p.w_hh[d];
    for (int idx = tid; idx < R * LDW; idx += 256) {
      const int r = idx / LDW, k = idx % LDW;
      Wl[idx] = (k < H) ? W[(long)((r & 3) * H + j0 + (r >> 2)) * H + k] : 0.f;
    }
    if (tid == 0) *abort_lds = 0;
  }

  const long hx_floats = (long)(H / 4) * Bp * 4;
  __amdgpu_buffer_rsrc_t hx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    hx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(p.hx + ((long)par * D + d) * hx_floats, 0, (int)(hx_floats * 4),
                                                     RSRC_FLAGS);
  unsigned* flags = p.flags + d * p.NC;

  // cell ownership: threads [0, MT*NT*64): tile (mt, nt), lane (n, q) -> unit 4*mt+q, batch 16*nt+n
  const bool owner = tid < MT * NT * 64;
  const int omt = owner ? (tid >> 6) / NT : 0, ont = owner ? (tid >> 6) % NT : 0;
  const int ob = 16 * ont + n, oj = j0 + 4 * omt + q;
  const int olen = (owner && ob < B) ? p.lens[ob] : 0;
  float c_state = 0.f;

  const int KG = (H + 15) / 16, KGW = (KG + 3) / 4;
  const int kg_begin = wave * KGW, kg_end = min(KG, kg_begin + KGW);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? s : T - 1 - s;
    // (1) this cell's gate pre-activations (independent of the recurrence: issue first)
    f32x4 xp = {0.f, 0.f, 0.f, 0.f};
    const long grow = ((long)t * B + ob) * D + d;
    if (owner && ob < B) xp = *reinterpret_cast<const f32x4*>(p.gates + grow * 4 * H + 4 * oj);

    // (2) h_{t-1} . W_hh^T for this slice, split-K over the four waves
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return;
      const __amdgpu_buffer_rsrc_t hr = hx_rsrc[(s - 1) & 1];
      for (int kg0 = kg_begin; kg0 < kg_end; kg0 += 8) {
        i32x4 hf[8][NT];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (kg0 + g < kg_end) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
              hf[g][j] = __builtin_amdgcn_raw_buffer_load_b128(hr, ((4 * (kg0 + g) + q) * Bp + 16 * j + n) * 16, 0, AUX_SC1);
          }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (kg0 + g < kg_end) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(&Wl[(16 * i + n) * LDW + 16 * (kg0 + g) + 4 * q]);
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const f32x4 h = __builtin_bit_cast(f32x4, hf[g][j]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], h[e], acc[i][j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    // (3) partial tiles -> LDS, reduce, gate math
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) part[((wave * MT + i) * NT + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    if (owner) {
      f32x4 g4 = xp;
#pragma unroll
      for (int w = 0; w < 4; ++w) g4 += part[((w * MT + omt) * NT + ont) * 64 + lane];
      const bool active = t < olen;
      float hval = 0.f;
      f32x4 gact = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
        c_state = fg * c_state + ig * gg;
        hval = og * tanhf(c_state);
        gact = (f32x4){ig, fg, gg, og};
      } else {
        c_state = 0.f;
      }
      // gather the 4 units of this tile row-group into lanes q == 0 (same wave: lanes n, n+16, n+32, n+48)
      f32x4 h4, c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h4[e] = __shfl(hval, n + 16 * e);
        c4[e] = __shfl(c_state, n + 16 * e);
      }
      if (q == 0) {  // exchange slice first: it is what the other workgroups wait for
        const int chunk = (j0 >> 2) + omt;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h4), hx_rsrc[s & 1], (chunk * Bp + ob) * 16, 0, AUX_SC1);
      }
      if (ob < B) {
        *reinterpret_cast<f32x4*>(p.gates + grow * 4 * H + 4 * oj) = gact;
        if (q == 0) {
          const int chunk = (j0 >> 2) + omt;
          *reinterpret_cast<f32x4*>(p.cst + ((((long)d * T + t) * (H / 4) + chunk) * B + ob) * 4) = c4;
          const long yo = grow * H + j0 + 4 * omt;
          *reinterpret_cast<f32x4*>(p.y + yo) = h4;
          if (p.ydrop) {
            f32x4 hd;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              hd[e] = (hash_u32(p.seed, (unsigned long long)(yo + e)) >= p.drop_thresh) ? h4[e] * p.keep_scale : 0.f;
            *reinterpret_cast<f32x4*>(p.ydrop + yo) = hd;
          }
        }
      }
    }
    // (4) publish step s
    publish_flag(flags + wg, (unsigned)(s + 1));
  }
}


// ------------------------------------------------------------------------------------------------
// backward recurrence
//   dh_t = dy_t + dG_{t'} . W_hh   (t' = the step processed just before), contraction over all 4H gate rows
//   v_mfma_f32_4x4x1_16B_f32: block = one hidden unit j' of a 16-unit group, A rows = this slice's 4 units,
//   B cols = 4 batch rows, 4 instructions (one per gate) per (16-unit group, batch quad).
// dynamic LDS: WT[UG][MT][64] f32x4 | part[16][MT*BG*16] floats | abort flag
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_ror_add(float x) {
  // sum over the 4 lanes of each 16-lane row that share lane%4 (row_ror:4, row_ror:8)
  int xi = __builtin_bit_cast(int, x);
  float y = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x124, 0xf, 0xf, false));
  int yi = __builtin_bit_cast(int, y);
  return y + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(yi, yi, 0x128, 0xf, 0xf, false));
}

template <int MT, int NT>
__global__ void __launch_bounds__(256) lstm_bwd_kernel(const LstmK p) {
  constexpr int BG = 4 * NT;  // batch quads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, B = p.B, D = p.D, T = p.T, Bp = p.Bp;
  const int UG = (H + 15) / 16, UGW = (UG + 3) / 4;
  f32x4* WT = reinterpret_cast<f32x4*>(smem);
  float* part = reinterpret_cast<float*>(WT + UG * MT * 64);
  int* abort_lds = reinterpret_cast<int*>(part + 16 * MT * BG * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 2, l4 = lane & 3, row = lane >> 4;
  const int d = blockIdx.x / p.NC, wg = blockIdx.x % p.NC;
  const int j0 = wg * p.Hs;

  // stage W_hh columns of this slice: WT[ug][m][lane = 4*blk + i][e] = W_hh[e*H + 16*ug + blk][j0 + 4*m + i]
  {
    const float* W = p.w_hh[d];
    for (int idx = tid; idx < UG * MT * 64; idx += 256) {
      const int ln = idx & 63, m = (idx >> 6) % MT, ug = (idx >> 6) / MT;
      const int jp = 16 * ug + (ln >> 2), jj = j0 + 4 * m + (ln & 3);
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (jp < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = W[(long)(e * H + jp) * H + jj];
      }
      WT[idx] = w;
    }
    if (tid == 0) *abort_lds = 0;
  }

  const long gx_floats = (long)H * Bp * 4;
  __amdgpu_buffer_rsrc_t gx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    gx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(p.hx + ((long)par * D + d) * gx_floats, 0, (int)(gx_floats * 4),
                                                     RSRC_FLAGS);
  unsigned* flags = p.flags + d * p.NC;

  // cell ownership: tid = ((m*BG + bg)*4 + i)*4 + jb  -> unit j0 + 4m + i, batch 4*bg + jb
  const bool owner = tid < MT * BG * 16;
  const int ojb = tid & 3, oi = (tid >> 2) & 3, obg = (tid >> 4) % BG, om = (tid >> 4) / BG;
  const int ob = 4 * obg + ojb, oj = j0 + 4 * om + oi;
  const int olen = (owner && ob < B) ? p.lens[ob] : 0;
  const int ochunk = oj >> 2;
  float dc_carry = 0.f;

  const int ug_begin = wave * UGW, ug_end = min(UG, ug_begin + UGW);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? T - 1 - s : s;
    const int tprev = (d == 0) ? t - 1 : t + 1;  // where c_{prev} of the forward recurrence lives
    // (1) prefetch this cell's stash (independent of the recurrence)
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
    float c_t = 0.f, c_p = 0.f, dyv = 0.f;
    const long grow = ((long)t * B + ob) * D + d;
    const bool active = owner && ob < B && t < olen;
    if (active) {
      gt = *reinterpret_cast<const f32x4*>(p.gates + grow * 4 * H + 4 * oj);
      c_t = p.cst[((((long)d * T + t) * (H / 4) + ochunk) * B + ob) * 4 + (oj & 3)];
      if (tprev >= 0 && tprev < T) c_p = p.cst[((((long)d * T + tprev) * (H / 4) + ochunk) * B + ob) * 4 + (oj & 3)];
      const long yo = grow * H + oj;
      dyv = p.dy[yo];
      if (p.ydrop) dyv = (hash_u32(p.seed, (unsigned long long)yo) >= p.drop_thresh) ? dyv * p.keep_scale : 0.f;
    }

    // (2) dG_{prev step} . W_hh[:, slice], split over unit groups (K) across the four waves
    f32x4 acc[MT][BG];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < BG; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return;
      const __amdgpu_buffer_rsrc_t gr = gx_rsrc[(s - 1) & 1];
      for (int ug = ug_begin; ug < ug_end; ++ug) {
        i32x4 gf[BG];
#pragma unroll
        for (int g = 0; g < BG; ++g)
          gf[g] = __builtin_amdgcn_raw_buffer_load_b128(gr, ((16 * ug + blk) * Bp + 4 * g + l4) * 16, 0, AUX_SC1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 w = WT[(ug * MT + m) * 64 + lane];
#pragma unroll
          for (int g = 0; g < BG; ++g) {
            const f32x4 x = __builtin_bit_cast(f32x4, gf[g]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[m][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[e], x[e], acc[m][g], 0, 0, 0);
          }
        }
      }
    }
    // (3) sum the 16 k-blocks: 4 lanes per row by DPP, then 4 rows x 4 waves through LDS
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < BG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = dpp_ror_add(acc[m][g][i]);
          if ((lane & 15) < 4) part[(wave * 4 + row) * (MT * BG * 16) + ((m * BG + g) * 4 + i) * 4 + l4] = v;
        }
    __syncthreads();
    if (owner) {
      float dh = dyv;
#pragma unroll
      for (int k = 0; k < 16; ++k) dh += part[k * (MT * BG * 16) + tid];
      f32x4 dg4 = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        const float ig = gt[0], fg = gt[1], gg = gt[2], og = gt[3];
        const float tc = tanhf(c_t);
        const float dc = dh * og * (1.f - tc * tc) + dc_carry;
        dg4[0] = dc * gg * ig * (1.f - ig);
        dg4[1] = dc * c_p * fg * (1.f - fg);
        dg4[2] = dc * ig * (1.f - gg * gg);
        dg4[3] = dh * tc * og * (1.f - og);
        dc_carry = dc * fg;
      } else {
        dc_carry = 0.f;
      }
      // exchange first (what the other workgroups wait for), then the stash for the weight-gradient GEMMs
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, dg4), gx_rsrc[s & 1], (oj * Bp + ob) * 16, 0, AUX_SC1);
      if (ob < B) *reinterpret_cast<f32x4*>(p.gates + grow * 4 * H + 4 * oj) = dg4;
    }
    // (4) publish step s
    publish_flag(flags + wg, (unsigned)(s + 1));
  }
}


// ================================================================================================
// v2 recurrence: D*G sync groups (direction x batch slice) of NC workgroups; a workgroup owns HS hidden units.
// Measured (tools/sync_probe.hip, profiles/r01_sync_probe.txt): one flag+gather round costs 1.8-2.7 us in a
// 32-member group with a 16 KB payload vs 6.5-9.4 us in a 128..256-member group with 64 KB -- the cost scales with
// members and bytes, not with XCD placement.  So the batch is cut into G slices of Bg <= 16 rows; each group's
// exchange vector is Bg x H (fwd) / Bg x 4H (bwd) and only NC = H/HS workgroups wait on each other.
// Matrix cores: v_mfma_f32_4x4x1_16B_f32 -- 16 independent 4x4 outer products per instruction.  Blocks are
// (k-slice, row-quad): fwd rows = the 4 gates of one unit, bwd rows = 4 units; columns = 4 batch rows.  No lane is
// wasted on an 8-row batch slice, and a lane ends up holding exactly the values one cell update needs.
// ================================================================================================

// acc[bq] += sum_{s < Ls} A(ks, s) (x) B(row, ks, s) over this wave's K range.  A comes from LDS (WA_w[s/4][lane] f32x4),
// B is gathered from the group's exchange rows (global, sc1) in 4 KB blocks through a wave-private LDS stage so that
// one 16-B ds_read broadcasts a batch row's values to the 16 lanes that need them.
template <int KS, int BQ>
__device__ __forceinline__ void gather_mma(__amdgpu_buffer_rsrc_t src, int row_stride_f, int kbase, int Ls,
                                           const f32x4* __restrict__ WA_w, float* __restrict__ hs, int lane, int myks,
                                           f32x4 (&acc)[BQ]) {
  constexpr int NB = 4 * BQ;             // batch rows of the group (padded)
  constexpr int SB = 256 / (NB * KS);    // 4-step columns per staged block (256 float4 = 4 KB)
  constexpr int LDB = 4 * SB + (SB > 1 ? 4 : 0);
  const int ns4 = Ls >> 2;
  const int nblk = (ns4 + SB - 1) / SB;
  f32x4 acc2[2][BQ];
#pragma unroll
  for (int bq = 0; bq < BQ; ++bq) acc2[0][bq] = acc2[1][bq] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int blk0 = 0; blk0 < nblk; blk0 += 4) {
    i32x4 r[4][4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = lane + 64 * q;
        const int s4l = f % SB, ks = (f / SB) % KS, row = f / (SB * KS);
        const int s4 = (blk0 + bb) * SB + s4l;
        const bool ok = (blk0 + bb) < nblk && s4 < ns4;
        const int off = ok ? (row * row_stride_f + kbase + ks * Ls + 4 * s4) * 4 : 0x7ffffff0;  // out of range reads 0
        r[bb][q] = __builtin_amdgcn_raw_buffer_load_b128(src, off, 0, AUX_SC1);
      }
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int blk = blk0 + bb;
      if (blk < nblk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = lane + 64 * q;
          const int s4l = f % SB, ks = (f / SB) % KS, row = f / (SB * KS);
          *reinterpret_cast<i32x4*>(&hs[(row * KS + ks) * LDB + 4 * s4l]) = r[bb][q];
        }
        const int nsl = min(SB, ns4 - blk * SB);
        // software pipeline: chunks of U steps, register ping-pong, so the next chunk's LDS reads are in flight
        // under this chunk's MFMAs; two accumulator sets (even/odd step) keep dependent MFMAs 2*BQ apart
        constexpr int U = 4;
        const f32x4* wa = WA_w + (long)blk * SB * 64 + lane;
        const float* hb = hs + ((lane & 3) * KS + myks) * LDB;
        auto lds_load = [&](int s0, f32x4(&av)[U], f32x4(&bv)[U][BQ]) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            av[u] = wa[(s0 + u) * 64];
#pragma unroll
            for (int bq = 0; bq < BQ; ++bq) bv[u][bq] = *reinterpret_cast<const f32x4*>(&hb[4 * bq * KS * LDB + 4 * (s0 + u)]);
          }
        };
        auto mma = [&](const f32x4(&av)[U], const f32x4(&bv)[U][BQ]) {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int bq = 0; bq < BQ; ++bq)
                acc2[u & 1][bq] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u][e], bv[u][bq][e], acc2[u & 1][bq], 0, 0, 0);
        };
        const int nfull = nsl / U;
        if (nfull > 0) {
          f32x4 a0[U], b0[U][BQ], a1[U], b1[U][BQ];
          lds_load(0, a0, b0);
          int c = 0;
          for (; c + 2 <= nfull; c += 2) {
            lds_load((c + 1) * U, a1, b1);
            mma(a0, b0);
            if (c + 2 < nfull) lds_load((c + 2) * U, a0, b0);
            mma(a1, b1);
          }
          if (c < nfull) mma(a0, b0);
        }
        for (int s4l = nfull * U; s4l < nsl; ++s4l) {
          const f32x4 av = wa[s4l * 64];
#pragma unroll
          for (int bq = 0; bq < BQ; ++bq) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&hb[4 * bq * KS * LDB + 4 * s4l]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[0][bq] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], bv[e], acc2[0][bq], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int bq = 0; bq < BQ; ++bq) acc[bq] += acc2[0][bq] + acc2[1][bq];
}

template <int KS, int BQ>
__host__ __device__ constexpr int stage_floats() {
  return 1024 + 4 * BQ * KS * 4;  // 4 KB block + per-(row, slice) padding
}

// dynamic LDS: WA[4][Ls/4][64] f32x4 | hs[4][stage_floats] | part[4][BQ][64] f32x4 | abort
// CELL: 0 LSTM (slots i,f,g,o), 1 GRU (slots r,z,n,-), 2 Elman RNN (slot 0; tanh or relu by p.cell).  Every cell type keeps
// the 4-slots-per-unit layout of the gate buffer and of the MFMA row-quads; unused slots carry zero weights.
template <int HS, int BQ, int CELL>
__global__ void __launch_bounds__(256) lstm_fwd2_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int KS = 16 / HS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, B = p.B, D = p.D, T = p.T, Kp = p.Kp;
  const int Kw = Kp / 4, Ls = Kw / KS, ns4 = Ls / 4;
  f32x4* WA = reinterpret_cast<f32x4*>(smem);
  float* hs_all = reinterpret_cast<float*>(WA + 4 * ns4 * 64);
  f32x4* part = reinterpret_cast<f32x4*>(hs_all + 4 * stage_floats<KS, BQ>());
  int* abort_lds = reinterpret_cast<int*>(part + 4 * BQ * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NG = D * p.G;
  // consecutive ids -> different groups; with a launch stride of 8 the XCD round-robin keeps every group on one XCD (speed only)
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;

  {  // weights: lane = 4*blk + i, blk = ks*HS + rq  <->  W_hh[(gate i)*H + j0 + rq][k], k = w*Kw + ks*Ls + 4*s4 + e
    const float* W = p.w_hh[d];
    for (int idx = tid; idx < 4 * ns4 * 64; idx += 256) {
      const int ln = idx & 63, s4 = (idx >> 6) % ns4, w = (idx >> 6) / ns4;
      const int i = ln & 3, blk = ln >> 2, rq = blk % HS, ks = blk / HS;
      const int k = w * Kw + ks * Ls + 4 * s4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < NGATE) {
        const float* row = W + (long)(i * H + j0 + rq) * H;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < H) v[e] = row[k + e];
      }
      WA[idx] = v;
    }
    if (tid == 0) *abort_lds = 0;
  }
  constexpr int NBR = 4 * BQ;
  const long hx_floats = (long)NBR * Kp;
  __amdgpu_buffer_rsrc_t hx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    hx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(p.hx + ((long)par * NG + gid) * hx_floats, 0, (int)(hx_floats * 4), RSRC_FLAGS);
  unsigned* flags = p.flags + gid * p.NC;

  // cell owners: tid = obq*(4*HS) + 4*unit + j : the lane that ends up with the 4 gate sums of (unit, batch row 4*obq+j)
  const bool owner = tid < BQ * 4 * HS;
  const int ol = tid % (4 * HS), obq = tid / (4 * HS);
  const int ounit = ol >> 2, brow = 4 * obq + (ol & 3);
  const int ob = b0 + brow, oj = j0 + ounit;
  const bool valid = owner && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float c_state = 0.f;  // LSTM: cell state; GRU / RNN: previous hidden state of this cell
  const float bhn = (CELL == 1 && owner) ? p.b_hh[d][2 * H + oj] : 0.f;
  const int myks = (lane >> 2) / HS;
  // per-step addresses advance by constant strides: keep running offsets instead of 64-bit multiplies in the loop
  const int t_first = (d == 0) ? 0 : T - 1;
  const long tdir = (d == 0) ? 1 : -1;
  long g_off = (((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj;                   // gates
  long c_off = ((((long)d * T + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4;        // cst
  long y_off = (((long)t_first * B + ob) * D + d) * H + oj;                           // y / y_drop
  const long g_step = tdir * (long)B * D * 4 * H, c_step = tdir * (long)H * B, y_step = tdir * (long)B * D * H;
  const int hx_off = (brow * Kp + oj) * 4;
  __syncthreads();
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? s : T - 1 - s;
    f32x4 xp = {0.f, 0.f, 0.f, 0.f};
    if (valid) xp = *reinterpret_cast<const f32x4*>(p.gates + g_off);

    f32x4 acc[BQ];
#pragma unroll
    for (int bq = 0; bq < BQ; ++bq) acc[bq] = (f32x4){0.f, 0.f, 0.f, 0.f};
    DBG_STAMP(0);  // prefetch issue
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return false;
      DBG_STAMP(1);  // flag wait
      gather_mma<KS, BQ>(hx_rsrc[(s - 1) & 1], Kp, wave * Kw, Ls, WA + wave * ns4 * 64,
                         hs_all + wave * stage_floats<KS, BQ>(), lane, myks, acc);
      DBG_STAMP(2);  // gather + MFMA
    }
#pragma unroll
    for (int bq = 0; bq < BQ; ++bq) part[(wave * BQ + bq) * 64 + lane] = acc[bq];
    __syncthreads();
    f32x4 gact = {0.f, 0.f, 0.f, 0.f}, h4 = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
    bool quad_lead = false;
    if (owner) {
      f32x4 rec = {0.f, 0.f, 0.f, 0.f};  // h_{t-1} . W_hh^T for this cell's slots
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rec += part[(w * BQ + obq) * 64 + ks * 4 * HS + ol];
      const bool active = valid && t < olen;
      float hval = 0.f;
      if (active) {
        if constexpr (CELL == 0) {
          const f32x4 g4 = xp + rec;
          const float ig = sig_sel(g4[0], p.hw_math), fg = sig_sel(g4[1], p.hw_math), gg = tanh_sel(g4[2], p.hw_math), og = sig_sel(g4[3], p.hw_math);
          c_state = fg * c_state + ig * gg;
          hval = og * tanh_sel(c_state, p.hw_math);
          gact = (f32x4){ig, fg, gg, og};
        } else if constexpr (CELL == 1) {
          const float rg = sig_sel(xp[0] + rec[0], p.hw_math), zg = sig_sel(xp[1] + rec[1], p.hw_math);
          const float hn = rec[2] + bhn;
          const float ng = tanh_sel(xp[2] + rg * hn, p.hw_math);
          hval = (1.f - zg) * ng + zg * c_state;
          c_state = hval;
          gact = (f32x4){rg, zg, ng, hn};
        } else {
          const float pre = xp[0] + rec[0];
          hval = (p.cell == RNNT_CELL_RNN_RELU) ? fmaxf(pre, 0.f) : tanh_sel(pre, p.hw_math);
          c_state = hval;
          gact = (f32x4){hval, 0.f, 0.f, 0.f};
        }
      } else {
        c_state = 0.f;
      }
      // 4 consecutive units of one batch row sit in lanes l, l+4, l+8, l+12: collect them in the first
      h4 = (f32x4){hval, row_shl<0x104>(hval), row_shl<0x108>(hval), row_shl<0x10C>(hval)};
      c4 = (f32x4){c_state, row_shl<0x104>(c_state), row_shl<0x108>(c_state), row_shl<0x10C>(c_state)};
      quad_lead = (ounit & 3) == 0;
      if (quad_lead)  // ONLY the exchange slice is stored before the flag: it is what the group waits for
        exchange_store<LOCAL>(__builtin_bit_cast(i32x4, h4), hx_rsrc[s & 1], hx_off);
    }
    DBG_STAMP(3);  // LDS reduce + cell math + exchange store issue
    publish_flag2<LOCAL>(flags + wg, (unsigned)(s + 1));
    DBG_STAMP(4);  // drain + barrier + flag
    // stash / layer-output stores AFTER the flag: they overlap the group's next exchange instead of delaying it
    if (valid) {
      *reinterpret_cast<f32x4*>(p.gates + g_off) = gact;
      if (quad_lead) {
        if constexpr (CELL == 0) *reinterpret_cast<f32x4*>(p.cst + c_off) = c4;
        const long yo = y_off;
        *reinterpret_cast<f32x4*>(p.y + yo) = h4;
        if (p.ydrop) {
          f32x4 hd;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            hd[e] = (hash_u32(p.seed, (unsigned long long)(yo + e)) >= p.drop_thresh) ? h4[e] * p.keep_scale : 0.f;
          *reinterpret_cast<f32x4*>(p.ydrop + yo) = hd;
        }
      }
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);  // stash stores issue
  }
  return true;
  };
  const bool ok = local ? run(std::true_type{}) : run(std::false_type{});
  if (!ok) return;
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

// ================================================================================================
// v3 forward: the recurrent product on the bf16 matrix cores, W_hh stationary in REGISTERS.
//
// Same decomposition as v2 (sync groups = direction x batch slice, a workgroup owns HS = 16 hidden units, split-K over the
// 4 waves), but h_{t-1} . W_hh^T runs as v_mfma_f32_16x16x32_bf16 on exact bf16 pieces (common.hpp: 6 piece products =
// fp32 accuracy) instead of v_mfma_f32_4x4x1_16B_f32: 96 MFMAs x 16 cycles per wave and step instead of 256 x 12.2.
//   M = the workgroup's 64 gate columns (4 blocks of 16 = 4 units x 4 gate slots), N = the group's batch rows (<= 16),
//   K = this wave's quarter of the hidden vector (NKS steps of 32).
//   A (stationary): lane -> gate column 16*mb + (lane&15), k = 32*ks + 8*(lane>>4) + e: the wave's W_hh slice as
//     3 bf16 pieces = 4*NKS*12 VGPRs, split once at kernel start: no LDS copy of the weights, no per-step weight reads.
//   B (per step): lane -> batch row (lane&15), the same 8 consecutive k.  The PRODUCER of a hidden value splits it (one cell
//     per lane: 4 VALU) and publishes three bf16 planes [row][Kp]; a consumer's operand piece is then 16 contiguous bytes:
//     three 16-B sc1 loads per k-step straight into MFMA operand registers, no LDS stage and no split on the critical path
//     (splitting at the consumers cost 176 VALU per lane and step in every one of the group's 32 workgroups).
//   D: lane -> (unit 4*mb + (lane>>4), row lane&15), its 4 registers = the 4 gate slots of that cell.
// Cross-wave reduction through LDS; wave w then owns units 4w..4w+3: one cell per lane, no DPP gymnastics.
// Stash layouts (gates, cst, y) are those of v2, so lstm_bwd2_kernel consumes them unchanged.
// dynamic LDS: part[4 waves][4 mb][64] f32x4 | abort
// ================================================================================================
// MB: 16-gate-column blocks per workgroup (the workgroup owns 4*MB hidden units); MB = 5 with 8 waves and K padded to 768 is
// the H = 640 form: 32 workgroups per sync group, which fit one XCD (40 would not)
// NLB: the last NLB of the MB gate-column blocks keep their W_hh pieces in LDS instead of registers (8-wave forms: two waves
// per SIMD share a 256-register cap)
template <int NKS, int CELL, int NWV = 4, int MB = 4, int NLB = 0>
__global__ void __launch_bounds__(64 * NWV) lstm_fwd3_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int HS = 4 * MB;
  static_assert(MB <= NWV, "one owner wave per gate-column block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, B = p.B, D = p.D, T = p.T, Kp = p.Kp;
  constexpr int Kw = 32 * NKS;
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  int* abort_lds = reinterpret_cast<int*>(part + NWV * MB * 64);
  constexpr int NRB = MB - NLB;
  u32x4* wl = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(abort_lds) + 16) + (long)(threadIdx.x >> 6) * NLB * NKS * 3 * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;
  const int lrow = lane & 15, lq = lane >> 4;

  bf16x8 wp[NRB][NKS][3];
  {
    const float* W = p.w_hh[d];
    const int gate = lrow & 3;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float* row = W + (long)(gate * H + j0 + 4 * mb + (lrow >> 2)) * H;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int k = wave * Kw + 32 * ks + 8 * lq;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (gate < NGATE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (k + e < H) lo[e] = row[k + e];
            if (k + 4 + e < H) hi[e] = row[k + 4 + e];
          }
        }
        if (mb < NRB) {
          split8(lo, hi, wp[mb < NRB ? mb : 0][ks]);
        } else {
          bf16x8 tmp[3];
          split8(lo, hi, tmp);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) wl[(((mb - NRB) * NKS + ks) * 3 + pl) * 64 + lane] = __builtin_bit_cast(u32x4, tmp[pl]);
        }
      }
    }
    if (tid == 0) *abort_lds = 0;
  }
  const int NBR = 4 * ((p.Bg + 3) / 4);  // exchange rows of the group (as allocated by the host: 4*BQ)
  // the exchange carries h already split by its producer: three bf16 planes [row][Kp] per (parity, group), 6 bytes per value
  const int plane_b = NBR * Kp * 2;
  const long hx_bytes = 3l * plane_b;
  __amdgpu_buffer_rsrc_t hx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    hx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.hx) + ((long)par * NG + gid) * hx_bytes, 0,
                                                     (int)hx_bytes, RSRC_FLAGS);
  unsigned* flags = p.flags + gid * p.NC;

  // one cell per lane of waves 0..3: unit 4*wave + lq of this workgroup, batch row lrow of this group (with NWV = 8 the
  // upper four waves only contribute their K-slice of the product)
  const bool ownw = wave < MB;
  const int ownb = ownw ? wave : 0;
  const int brow = lrow;
  const int ob = b0 + brow, oj = j0 + 4 * ownb + lq;
  const bool inrow = ownw && brow < NBR;
  const bool valid = ownw && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float c_state = 0.f;
  const float bhn = (CELL == 1 && ownw) ? p.b_hh[d][2 * H + oj] : 0.f;
  const int t_first = (d == 0) ? 0 : T - 1;
  const long tdir = (d == 0) ? 1 : -1;
  long g_off = (((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj;
  long c_off = ((((long)d * T + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3);
  long y_off = (((long)t_first * B + ob) * D + d) * H + oj;
  const long g_step = tdir * (long)B * D * 4 * H, c_step = tdir * (long)H * B, y_step = tdir * (long)B * D * H;
  const int hx_off = (brow * Kp + oj) * 2;
  const int gat_off = brow < NBR ? (brow * Kp + wave * Kw + 8 * lq) * 2 : 0x7ffffff0;  // rows beyond the group read 0
  __syncthreads();
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? s : T - 1 - s;
    f32x4 xp = {0.f, 0.f, 0.f, 0.f};
    if (valid) xp = *reinterpret_cast<const f32x4*>(p.gates + g_off);

    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    DBG_STAMP(0);
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return false;
      DBG_STAMP(1);
      i32x4 raw[NKS][3];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          raw[ks][pl] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc[(s - 1) & 1], gat_off + 64 * ks + pl * plane_b, 0, AUX_SC1);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        bf16x8 hp[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) hp[pl] = __builtin_bit_cast(bf16x8, raw[ks][pl]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          bf16x8 w0, w1, w2;
          if (mb < NRB) {
            w0 = wp[mb < NRB ? mb : 0][ks][0]; w1 = wp[mb < NRB ? mb : 0][ks][1]; w2 = wp[mb < NRB ? mb : 0][ks][2];
          } else {
            const u32x4* f = wl + ((mb - NRB) * NKS + ks) * 3 * 64 + lane;
            w0 = __builtin_bit_cast(bf16x8, f[0]); w1 = __builtin_bit_cast(bf16x8, f[64]); w2 = __builtin_bit_cast(bf16x8, f[128]);
          }
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, hp[0], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, hp[1], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, hp[2], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, hp[0], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, hp[1], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, hp[0], acc[mb], 0, 0, 0);
        }
      }
      DBG_STAMP(2);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) part[(wave * MB + mb) * 64 + lane] = acc[mb];
    __syncthreads();
    f32x4 rec = part[ownb * 64 + lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w) rec += part[(w * MB + ownb) * 64 + lane];
    const bool active = valid && t < olen;
    float hval = 0.f;
    f32x4 gact = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      if constexpr (CELL == 0) {
        const f32x4 g4 = xp + rec;
        const float ig = sig_sel(g4[0], p.hw_math), fg = sig_sel(g4[1], p.hw_math), gg = tanh_sel(g4[2], p.hw_math), og = sig_sel(g4[3], p.hw_math);
        c_state = fg * c_state + ig * gg;
        hval = og * tanh_sel(c_state, p.hw_math);
        gact = (f32x4){ig, fg, gg, og};
      } else if constexpr (CELL == 1) {
        const float rg = sig_sel(xp[0] + rec[0], p.hw_math), zg = sig_sel(xp[1] + rec[1], p.hw_math);
        const float hn = rec[2] + bhn;
        const float ng = tanh_sel(xp[2] + rg * hn, p.hw_math);
        hval = (1.f - zg) * ng + zg * c_state;
        c_state = hval;
        gact = (f32x4){rg, zg, ng, hn};
      } else {
        const float pre = xp[0] + rec[0];
        hval = (p.cell == RNNT_CELL_RNN_RELU) ? fmaxf(pre, 0.f) : tanh_sel(pre, p.hw_math);
        c_state = hval;
        gact = (f32x4){hval, 0.f, 0.f, 0.f};
      }
    } else {
      c_state = 0.f;
    }
    if (inrow) {  // ONLY the exchange slice is stored before the flag: h = h0 + h1 + h2 exactly, one bf16 per plane
      const unsigned u0 = __float_as_uint(hval);
      const float r1 = hval - __uint_as_float(u0 & 0xffff0000u);
      const unsigned u1 = __float_as_uint(r1);
      const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
      constexpr int AUX = LOCAL ? 0 : AUX_SC1;
      __builtin_amdgcn_raw_buffer_store_b16((short)(u0 >> 16), hx_rsrc[s & 1], hx_off, 0, AUX);
      __builtin_amdgcn_raw_buffer_store_b16((short)(u1 >> 16), hx_rsrc[s & 1], hx_off + plane_b, 0, AUX);
      __builtin_amdgcn_raw_buffer_store_b16((short)(u2 >> 16), hx_rsrc[s & 1], hx_off + 2 * plane_b, 0, AUX);
    }
    DBG_STAMP(3);
    publish_flag2<LOCAL>(flags + wg, (unsigned)(s + 1));
    DBG_STAMP(4);
    if (valid) {
      *reinterpret_cast<f32x4*>(p.gates + g_off) = gact;
      if constexpr (CELL == 0) p.cst[c_off] = c_state;
      p.y[y_off] = hval;
      if (p.ydrop)
        p.ydrop[y_off] = (hash_u32(p.seed, (unsigned long long)y_off) >= p.drop_thresh) ? hval * p.keep_scale : 0.f;
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);
  }
  return true;
  };
  const bool ok = local ? run(std::true_type{}) : run(std::false_type{});
  if (!ok) return;
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

// dynamic LDS: WA[4][Ls/4][64] f32x4 | hs[4][stage_floats] | part[4][BQ][64] f32x4 | abort
template <int HS, int BQ, int CELL>
__global__ void __launch_bounds__(256) lstm_bwd2_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int UQ = HS / 4, KS = 16 / UQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, B = p.B, D = p.D, T = p.T, Kp = p.Kp;
  const int Kw = Kp, Ls = Kw / KS, ns4 = Ls / 4;  // contraction index k = 4*j' + gate over 4*Kp, a quarter per wave
  f32x4* WA = reinterpret_cast<f32x4*>(smem);
  float* hs_all = reinterpret_cast<float*>(WA + 4 * ns4 * 64);
  float* part = hs_all + 4 * stage_floats<KS, BQ>();
  int* abort_lds = reinterpret_cast<int*>(part + 4 * BQ * 64 * 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;

  {  // weights: lane = 4*blk + i, blk = ks*UQ + uq  <->  W_hh[(gate k&3)*H + (k>>2)][j0 + 4*uq + i]
    const float* W = p.w_hh[d];
    for (int idx = tid; idx < 4 * ns4 * 64; idx += 256) {
      const int ln = idx & 63, s4 = (idx >> 6) % ns4, w = (idx >> 6) / ns4;
      const int i = ln & 3, blk = ln >> 2, uq = blk % UQ, ks = blk / UQ;
      const int k = w * Kw + ks * Ls + 4 * s4;  // multiple of 4: e is the gate, k>>2 the unit j'
      const int jp = k >> 2, col = j0 + 4 * uq + i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (jp < H) {
#pragma unroll
        for (int e = 0; e < NGATE; ++e) v[e] = W[(long)(e * H + jp) * H + col];
      }
      WA[idx] = v;
    }
    if (tid == 0) *abort_lds = 0;
  }
  constexpr int NBR = 4 * BQ;
  const long gx_floats = (long)NBR * 4 * Kp;
  __amdgpu_buffer_rsrc_t gx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    gx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(p.hx + ((long)par * NG + gid) * gx_floats, 0, (int)(gx_floats * 4), RSRC_FLAGS);
  unsigned* flags = p.flags + gid * p.NC;

  // cell owners: tid = ((obq*UQ + uq)*4 + i)*4 + j -> unit j0 + 4*uq + i, batch row 4*obq + j
  const bool owner = tid < BQ * HS * 4;
  const int ojb = tid & 3, oi = (tid >> 2) & 3, ouq = (tid >> 4) % UQ, obq = (tid >> 4) / UQ;
  const int brow = 4 * obq + ojb, ob = b0 + brow, oj = j0 + 4 * ouq + oi;
  const bool valid = owner && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float dc_carry = 0.f;
  const int myks = (lane >> 2) / UQ;
  const int t_first = (d == 0) ? T - 1 : 0;
  const long tdir = (d == 0) ? -1 : 1;
  long g_off = (((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj;
  long c_off = ((((long)d * T + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3);
  long y_off = (((long)t_first * B + ob) * D + d) * H + oj;
  const long g_step = tdir * (long)B * D * 4 * H, c_step = tdir * (long)H * B, y_step = tdir * (long)B * D * H;
  const int gx_off = (brow * 4 * Kp + 4 * oj) * 4;
  __syncthreads();
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? T - 1 - s : s;
    const int tprev = (d == 0) ? t - 1 : t + 1;
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
    float c_t = 0.f, c_p = 0.f, dyv = 0.f;
    const bool active = valid && t < olen;
    if (active) {
      gt = *reinterpret_cast<const f32x4*>(p.gates + g_off);
      if constexpr (CELL == 0) {
        c_t = p.cst[c_off];
        if (tprev >= 0 && tprev < T) c_p = p.cst[c_off + c_step];  // the backward walks towards the forward's t_prev
      } else if constexpr (CELL == 1) {
        if (tprev >= 0 && tprev < T) c_p = p.y[y_off + y_step];    // GRU: h_{t_prev} of this cell (0 at the sequence start)
      }
      const long yo = y_off;
      dyv = p.dy[yo];
      if (p.ydrop) dyv = (hash_u32(p.seed, (unsigned long long)yo) >= p.drop_thresh) ? dyv * p.keep_scale : 0.f;
    }

    f32x4 acc[BQ];
#pragma unroll
    for (int bq = 0; bq < BQ; ++bq) acc[bq] = (f32x4){0.f, 0.f, 0.f, 0.f};
    DBG_STAMP(0);  // prefetch issue
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return false;
      DBG_STAMP(1);  // flag wait
      gather_mma<KS, BQ>(gx_rsrc[(s - 1) & 1], 4 * Kp, wave * Kw, Ls, WA + wave * ns4 * 64,
                         hs_all + wave * stage_floats<KS, BQ>(), lane, myks, acc);
      DBG_STAMP(2);  // gather + MFMA
    }
#pragma unroll
    for (int bq = 0; bq < BQ; ++bq) *reinterpret_cast<f32x4*>(&part[((wave * BQ + bq) * 64 + lane) * 4]) = acc[bq];
    __syncthreads();
    f32x4 dg4 = {0.f, 0.f, 0.f, 0.f}, dgh4 = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      float dh = dyv;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dh += part[((w * BQ + obq) * 64 + 4 * (ks * UQ + ouq) + ojb) * 4 + oi];
      if (active) {
        if constexpr (CELL == 0) {
          const float ig = gt[0], fg = gt[1], gg = gt[2], og = gt[3];
          const float tc = tanh_sel(c_t, p.hw_math);
          const float dc = dh * og * (1.f - tc * tc) + dc_carry;
          dg4[0] = dc * gg * ig * (1.f - ig);
          dg4[1] = dc * c_p * fg * (1.f - fg);
          dg4[2] = dc * ig * (1.f - gg * gg);
          dg4[3] = dh * tc * og * (1.f - og);
          dc_carry = dc * fg;
          dgh4 = dg4;
        } else if constexpr (CELL == 1) {
          const float rg = gt[0], zg = gt[1], ng = gt[2], hn = gt[3];
          dh += dc_carry;                       // direct path dh_t -> dh_{t_prev} through z
          const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
          const float dz_pre = dh * (c_p - ng) * zg * (1.f - zg);
          const float dr_pre = dn_pre * hn * rg * (1.f - rg);
          dg4 = (f32x4){dr_pre, dz_pre, dn_pre, 0.f};        // input side: dX, dW_ih, db_ih
          dgh4 = (f32x4){dr_pre, dz_pre, dn_pre * rg, 0.f};  // hidden side: W_hh^T product, dW_hh, db_hh
          dc_carry = dh * zg;
        } else {
          const float hv = gt[0];
          const float dpre = (p.cell == RNNT_CELL_RNN_RELU) ? (hv > 0.f ? dh : 0.f) : dh * (1.f - hv * hv);
          dg4 = (f32x4){dpre, 0.f, 0.f, 0.f};
          dgh4 = dg4;
        }
      } else {
        dc_carry = 0.f;
      }
      exchange_store<LOCAL>(__builtin_bit_cast(i32x4, dgh4), gx_rsrc[s & 1], gx_off);
    }
    DBG_STAMP(3);  // LDS reduce + cell math + exchange store issue
    publish_flag2<LOCAL>(flags + wg, (unsigned)(s + 1));
    DBG_STAMP(4);  // drain + barrier + flag
    if (valid) {  // stash after the flag (off the critical path)
      *reinterpret_cast<f32x4*>(p.gates + g_off) = dg4;
      if constexpr (CELL == 1) *reinterpret_cast<f32x4*>(p.aux + g_off) = dgh4;
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);  // stash stores issue
  }
  return true;
  };
  const bool ok = local ? run(std::true_type{}) : run(std::false_type{});
  if (!ok) return;
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

// ================================================================================================
// v4 backward ("scatter form"): the recurrent product dh_{t-1} = dG_t . W_hh with the roles of the exchange swapped.
//
// v2 gathers ALL of dG_t (rows x 4H: 64 KB per workgroup and step at H = 512) and multiplies by its 16 columns of W_hh.
// Here a workgroup multiplies ITS OWN 64 gate columns of dG_t (rows x 64, local, produced by its own cell math) by its 64
// ROWS of W_hh and publishes the partial dh_{t-1} for ALL hidden units (rows x H); a consumer then sums, for its 16 units,
// the NC partial slices (NC x rows x 16 floats = 16 KB at H = 512: the forward's exchange volume, a quarter of v2's).
// The product runs on the bf16 matrix cores on exact bf16 pieces (common.hpp), W_hh stationary in registers:
//   M = output units (this wave's Kp/64 blocks of 16), N = the group's batch rows (<= 16), K = the 64 own gate columns
//   (2 steps of 32);  A: lane -> output unit 16*mb + (lane&15), k = 32*ks + 8*(lane>>4) + e = 4*(own unit) + gate;
//   B: lane -> row (lane&15), the same 8 gate columns, read from a 4 KB LDS image of the cell math's dG and split;
//   D: lane -> (units 16*mb + 4*(lane>>4) .. +3, row lane&15): one 16-byte store straight into the exchange buffer.
// Exchange buffer: partial[parity][group][producer][row][Kp] fp32.  Protocol, owners, cell math and stash as in v2.
// dynamic LDS: red[256] f32x4 | dgs[16][DGS_LD] float | abort
// ================================================================================================
// NOB: 16-unit output blocks per wave (Kp / 16 / NWV); MB: the workgroup owns 4*MB units = 16*MB gate columns
// NLB: of a wave's NOB output blocks, the last NLB keep their W_hh pieces in LDS (MFMA-operand order, one 16-byte fragment
// per lane) instead of registers: the 8-wave forms run two waves per SIMD under a 256-register cap
template <int NOB, int BQ, int CELL, int NWV = 4, int MB = 4, int NLB = 0>
__global__ void __launch_bounds__(64 * NWV) lstm_bwd4_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int HS = 4 * MB, UQ = MB;
  constexpr int NT = 64 * NWV;
  constexpr int NMB = NOB;
  constexpr int NRB = NOB - NLB;          // output blocks whose W pieces stay in registers
  constexpr int KSB = (16 * MB + 31) / 32;   // 32-deep k-steps over the own gate columns (zero padded)
  constexpr int NBR = 4 * BQ;           // exchange rows of the group
  constexpr int DGS_LD = 32 * KSB + 4;   // floats per row of the dG image (+ pad)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, B = p.B, D = p.D, T = p.T, Kp = p.Kp;
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* dgs = reinterpret_cast<float*>(red + NT);
  int* abort_lds = reinterpret_cast<int*>(dgs + 16 * DGS_LD);
  // [wave][NLB][KSB][3][64] 16-byte fragments, after a 16-byte aligned gap
  u32x4* wl = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(abort_lds) + 16) + (long)(threadIdx.x >> 6) * NLB * KSB * 3 * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;
  const int lrow = lane & 15, lq = lane >> 4;

  bf16x8 wp[NRB > 0 ? NRB : 1][KSB][3];
  {
    const float* W = p.w_hh[d];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int u = 16 * (wave * NMB + mb) + lrow;  // output unit = column of W_hh
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const int c = 32 * ks + 8 * lq;             // own gate column 4*unit + gate; c is a multiple of 8
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (u < H) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (e < NGATE) {
              if ((c >> 2) < HS) lo[e] = W[(long)(e * H + j0 + (c >> 2)) * H + u];
              if ((c >> 2) + 1 < HS) hi[e] = W[(long)(e * H + j0 + (c >> 2) + 1) * H + u];
            }
          }
        }
        if (mb < NRB) {
          split8(lo, hi, wp[mb < NRB ? mb : 0][ks]);
        } else {
          bf16x8 tmp[3];
          split8(lo, hi, tmp);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) wl[(((mb - NRB) * KSB + ks) * 3 + pl) * 64 + lane] = __builtin_bit_cast(u32x4, tmp[pl]);
        }
      }
    }
    for (int i = tid; i < 16 * DGS_LD; i += NT) dgs[i] = 0.f;
    if (tid == 0) *abort_lds = 0;
  }
  const long px_floats = (long)p.NC * NBR * Kp;  // one group's partials of one parity
  __amdgpu_buffer_rsrc_t px_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    px_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(p.hx + ((long)par * NG + gid) * px_floats, 0, (int)(px_floats * 4), RSRC_FLAGS);
  unsigned* flags = p.flags + gid * p.NC;

  // cell owners (as v2): tid = ((obq*UQ + uq)*4 + i)*4 + j -> unit j0 + 4*uq + i, batch row 4*obq + j
  const bool owner = tid < BQ * HS * 4;
  const int ojb = tid & 3, oi = (tid >> 2) & 3, ouq = (tid >> 4) % UQ, obq = (tid >> 4) / UQ;  // valid for tid < BQ*HS*4
  const int brow = 4 * obq + ojb, ob = b0 + brow, oj = j0 + 4 * ouq + oi;
  const bool valid = owner && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float dc_carry = 0.f;
  f32x4 db_acc = {0.f, 0.f, 0.f, 0.f}, dbh_acc = {0.f, 0.f, 0.f, 0.f};  // bias gradients: time sums of this cell's dG
  const int t_first = (d == 0) ? T - 1 : 0;
  const long tdir = (d == 0) ? -1 : 1;
  long g_off = (((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj;
  long c_off = ((((long)d * T + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3);
  long y_off = (((long)t_first * B + ob) * D + d) * H + oj;
  const long g_step = tdir * (long)B * D * 4 * H, c_step = tdir * (long)H * B, y_step = tdir * (long)B * D * H;
  // gather: thread -> (row, unit quad) pair pr = tid % (NBR*UQ) and producer class q = tid / (NBR*UQ); it sums the partial
  // slices of producers q, q + NQ, ... for that pair; owners then add the NQ class sums
  constexpr int NPAIR = NBR * UQ, NQ = NT / NPAIR;
  const int gq = tid / NPAIR, gpr = tid % NPAIR;
  const int grow = gpr / UQ, guq = gpr % UQ;
  const bool gact = gq < NQ;
  const int gat_base = (gq * NBR + grow) * Kp + j0 + 4 * guq;  // floats; + NQ producers per visit
  const int gat_step = NQ * NBR * Kp;
  // publish: lane -> row lrow, units 16*mbg + 4*lq .. +3
  const int pub_base = lrow < NBR ? ((wg * NBR + lrow) * Kp + 16 * wave * NMB + 4 * lq) * 4 : 0x7ffffff0;
  __syncthreads();
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? T - 1 - s : s;
    const int tprev = (d == 0) ? t - 1 : t + 1;
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
    float c_t = 0.f, c_p = 0.f, dyv = 0.f;
    const bool active = valid && t < olen;
    if (active) {
      gt = *reinterpret_cast<const f32x4*>(p.gates + g_off);
      if constexpr (CELL == 0) {
        c_t = p.cst[c_off];
        if (tprev >= 0 && tprev < T) c_p = p.cst[c_off + c_step];
      } else if constexpr (CELL == 1) {
        if (tprev >= 0 && tprev < T) c_p = p.y[y_off + y_step];
      }
      const long yo = y_off;
      dyv = p.dy[yo];
      if (p.ydrop) dyv = (hash_u32(p.seed, (unsigned long long)yo) >= p.drop_thresh) ? dyv * p.keep_scale : 0.f;
    }
    DBG_STAMP(0);  // prefetch issue
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (!wait_flags(flags, p.NC, (unsigned)s, p.status, abort_lds)) return false;
      DBG_STAMP(1);  // flag wait
      const __amdgpu_buffer_rsrc_t src = px_rsrc[(s - 1) & 1];
      constexpr int NI = 4;
      for (int w0 = 0; w0 < p.NC; w0 += NQ * NI) {
        i32x4 r[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const bool ok = gact && w0 + gq + NQ * i < p.NC;
          r[i] = __builtin_amdgcn_raw_buffer_load_b128(src, ok ? (gat_base + (w0 / NQ + i) * gat_step) * 4 : 0x7ffffff0, 0, AUX_SC1);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) gsum += __builtin_bit_cast(f32x4, r[i]);
      }
    }
    red[tid] = gsum;
    __syncthreads();
    DBG_STAMP(2);  // gather + partial sums
    f32x4 dg4 = {0.f, 0.f, 0.f, 0.f}, dgh4 = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      float dh = dyv;
#pragma unroll
      for (int q = 0; q < NQ; ++q) dh += red[q * NPAIR + brow * UQ + ouq][oi];
      if (active) {
        if constexpr (CELL == 0) {
          const float ig = gt[0], fg = gt[1], gg = gt[2], og = gt[3];
          const float tc = tanh_sel(c_t, p.hw_math);
          const float dc = dh * og * (1.f - tc * tc) + dc_carry;
          dg4[0] = dc * gg * ig * (1.f - ig);
          dg4[1] = dc * c_p * fg * (1.f - fg);
          dg4[2] = dc * ig * (1.f - gg * gg);
          dg4[3] = dh * tc * og * (1.f - og);
          dc_carry = dc * fg;
          dgh4 = dg4;
        } else if constexpr (CELL == 1) {
          const float rg = gt[0], zg = gt[1], ng = gt[2], hn = gt[3];
          dh += dc_carry;
          const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
          const float dz_pre = dh * (c_p - ng) * zg * (1.f - zg);
          const float dr_pre = dn_pre * hn * rg * (1.f - rg);
          dg4 = (f32x4){dr_pre, dz_pre, dn_pre, 0.f};
          dgh4 = (f32x4){dr_pre, dz_pre, dn_pre * rg, 0.f};
          dc_carry = dh * zg;
        } else {
          const float hv = gt[0];
          const float dpre = (p.cell == RNNT_CELL_RNN_RELU) ? (hv > 0.f ? dh : 0.f) : dh * (1.f - hv * hv);
          dg4 = (f32x4){dpre, 0.f, 0.f, 0.f};
          dgh4 = dg4;
        }
      } else {
        dc_carry = 0.f;
      }
      db_acc += dg4;
      if constexpr (CELL == 1) dbh_acc += dgh4;
      *reinterpret_cast<f32x4*>(&dgs[brow * DGS_LD + 4 * (4 * ouq + oi)]) = dgh4;  // own gate column 4*unit + gate
    }
    __syncthreads();
    DBG_STAMP(3);  // LDS reduce + cell math
    {
      bf16x8 gp[KSB][3];
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const float* src = dgs + lrow * DGS_LD + 32 * ks + 8 * lq;
        split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), gp[ks]);
      }
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
          bf16x8 w0, w1, w2;
          if (mb < NRB) {
            w0 = wp[mb < NRB ? mb : 0][ks][0]; w1 = wp[mb < NRB ? mb : 0][ks][1]; w2 = wp[mb < NRB ? mb : 0][ks][2];
          } else {
            const u32x4* f = wl + ((mb - NRB) * KSB + ks) * 3 * 64 + lane;
            w0 = __builtin_bit_cast(bf16x8, f[0]); w1 = __builtin_bit_cast(bf16x8, f[64]); w2 = __builtin_bit_cast(bf16x8, f[128]);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, gp[ks][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, gp[ks][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, gp[ks][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, gp[ks][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, gp[ks][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, gp[ks][0], acc, 0, 0, 0);
        }
        exchange_store<LOCAL>(__builtin_bit_cast(i32x4, acc), px_rsrc[s & 1], pub_base + 64 * mb);
      }
    }
    publish_flag2<LOCAL>(flags + wg, (unsigned)(s + 1));
    DBG_STAMP(4);  // MFMA + partial stores + drain + barrier + flag
    if (valid) {
      *reinterpret_cast<f32x4*>(p.gates + g_off) = dg4;
      if constexpr (CELL == 1) *reinterpret_cast<f32x4*>(p.aux + g_off) = dgh4;
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);  // stash stores issue
  }
  return true;
  };
  const bool ok = local ? run(std::true_type{}) : run(std::false_type{});
  if (!ok) return;
  if (owner) {  // one row of the (group, exchange row) table per cell row; summed over rows and groups by db_reduce_kernel
    const long row = (long)gid * NBR + brow;
    *reinterpret_cast<f32x4*>(p.dbp + row * 4 * H + 4 * oj) = db_acc;
    if constexpr (CELL == 1) *reinterpret_cast<f32x4*>(p.dbhp + row * 4 * H + 4 * oj) = dbh_acc;
  }
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// db[d][g*H + j] = sum over the direction's (group, row) table of part[(d*rows_per_dir + r)*4H + 4j+g], fixed order
__global__ void db_reduce_kernel(const float* __restrict__ part, int rows_per_dir, int H, int ngate, float* __restrict__ o0,
                                 float* __restrict__ o1, int accumulate, float* __restrict__ p0 = nullptr,
                                 float* __restrict__ p1 = nullptr) {
  const int c = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
  if (c >= 4 * H || (c & 3) >= ngate) return;
  const float* src = part + (long)d * rows_per_dir * 4 * H + c;
  float s = 0.f;
  for (int r = 0; r < rows_per_dir; ++r) s += src[(long)r * 4 * H];
  float* o = (d ? o1 : o0) + (c & 3) * H + (c >> 2);
  *o = accumulate ? *o + s : s;
  float* q = d ? p1 : p0;  // optional second destination (LSTM / Elman: grad b_hh == grad b_ih)
  if (q) {
    q += (c & 3) * H + (c >> 2);
    *q = accumulate ? *q + s : s;
  }
}
// out[(d*4H + 4j+g)*I + k] = g < ngate ? w[d][(g*H + j)*I + k] : 0      (4 slots per unit whatever the cell type)
__global__ void permute_w_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int H, int I, int ngate,
                                 float* __restrict__ out) {
  const long per = (long)4 * H * I;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int d = blockIdx.y;
  if (idx >= per) return;
  const int k = (int)(idx % I), r = (int)(idx / I);
  const float* w = d ? w1 : w0;
  out[d * per + idx] = (r & 3) < ngate ? w[(long)((r & 3) * H + (r >> 2)) * I + k] : 0.f;
}
// inverse for gradients: dw[d][(g*H + j)*I + k] = in[(d*4H + 4j+g)*I + k]
__global__ void unpermute_w_kernel(const float* __restrict__ in, int H, int I, long in_dir_stride, int ngate,
                                   float* __restrict__ o0, float* __restrict__ o1, int accumulate, float* __restrict__ p0 = nullptr,
                                   float* __restrict__ p1 = nullptr) {
  const long per = (long)4 * H * I;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int d = blockIdx.y;
  if (idx >= per) return;
  const int k = (int)(idx % I), r = (int)(idx / I);
  float* o = d ? o1 : o0;
  if ((r & 3) < ngate) {
    const long off = (long)((r & 3) * H + (r >> 2)) * I + k;
    const float v = in[d * in_dir_stride + idx];
    o[off] = accumulate ? o[off] + v : v;
    float* q = d ? p1 : p0;
    if (q) q[off] = accumulate ? q[off] + v : v;
  }
}
// bias folded into the hoisted input projection: b_ih + b_hh per slot; GRU keeps b_hn out (it sits inside r * (.))
__global__ void permute_bias_kernel(const float* __restrict__ bi0, const float* __restrict__ bh0,
                                    const float* __restrict__ bi1, const float* __restrict__ bh1, int H, int ngate,
                                    int gru, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int d = blockIdx.y;
  if (idx >= 4 * H) return;
  const int g = idx & 3, src = g * H + (idx >> 2);
  float v = 0.f;
  if (g < ngate) {
    const float bi = d ? bi1[src] : bi0[src], bh = d ? bh1[src] : bh0[src];
    v = (gru && g == 2) ? bi : bi + bh;
  }
  out[d * 4 * H + idx] = v;
}

// two-stage deterministic column sum.  stage 1: grid (ceil(N/64), RC): block (64 columns x 4 row lanes) sums its
// row chunk into part[rc][n]; stage 2: out[n] = sum_rc part[rc][n] in fixed order.
constexpr int COLSUM_RC_MAX = 128;
inline int colsum_chunks(long M, long N) {
  long want = ceil_div(2048, ceil_div(N, 64));  // ~2048 blocks in flight
  const long by_m = ceil_div(M, 64);
  if (want > by_m) want = by_m;
  if (want > COLSUM_RC_MAX) want = COLSUM_RC_MAX;
  return (int)(want < 1 ? 1 : want);
}
__global__ void __launch_bounds__(256) colsum_stage1_kernel(const float* __restrict__ X, long M, long N, long ld,
                                                            long rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 64 + c;
  const long m0 = (long)blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f;
  if (n < N)
    for (long m = m0 + r; m < m1; m += 4) s += X[m * ld + n];
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && n < N) part[(long)blockIdx.y * N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void __launch_bounds__(256) colsum_stage2_kernel(const float* __restrict__ part, long N, int rc,
                                                            float* __restrict__ out, int accumulate) {
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < rc; ++k) s += part[(long)k * N + n];
  out[n] = accumulate ? out[n] + s : s;
}
int launch_colsum(const float* X, long M, long N, long ld, float* out, void* ws, size_t ws_bytes, hipStream_t s,
                  int accumulate = 0) {
  const int rc = colsum_chunks(M, N);
  RNNT_CHECK_ARG(ws && ws_bytes >= (size_t)rc * N * 4, "colsum: workspace too small (%zu < %zu)", ws_bytes, (size_t)rc * N * 4);
  ProfScope prof(RNNT_K_MISC, 4.0 * (double)M * (double)N, s);
  const long rows = ceil_div(M, rc);
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((unsigned)ceil_div(N, 64), rc), dim3(256), 0, s, X, M, N, ld, rows, (float*)ws);
  RNNT_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, (const float*)ws, N, rc, out, accumulate);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

__global__ void embedding_fwd_kernel(const float* __restrict__ W, const long* __restrict__ idx, long M, int H, int V,
                                     float* __restrict__ out) {
  const long total = M * H;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / H;
    const int h = (int)(i % H);
    const long v = idx[m];
    out[i] = (v >= 0 && v < V) ? W[v * H + h] : 0.f;
  }
}

__global__ void __launch_bounds__(256) embedding_bwd_kernel(const float* __restrict__ dE, const long* __restrict__ idx, long M, int H, int V,
                                                            long pad, float* __restrict__ dW, int accumulate) {
  // one workgroup per vocabulary row: the tokens that hit it are compacted IN ORDER (ballot prefix) into LDS, then summed in that
  // fixed order (deterministic, no atomics) — the scan over all M tokens is done once per row, not once per feature
  constexpr int CAP = 2048;
  __shared__ int list[CAP];
  __shared__ int wcnt[4], nlist;
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (v == pad) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // features tid, tid + 256, ... (H <= 1024 keeps everything in registers; more: extra passes)
  for (int h0 = 0; h0 < H; h0 += 1024) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = 0.f;
    for (long mb = 0; mb < M; mb += CAP) {   // batches of CAP tokens
      if (tid == 0) nlist = 0;
      __syncthreads();
      const long mend = min(M, mb + CAP);
      for (long m0 = mb; m0 < mend; m0 += 256) {
        const long m = m0 + tid;
        const bool hit = m < mend && idx[m] == v;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int base = nlist;
        for (int w = 0; w < wave; ++w) base += wcnt[w];
        if (hit) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(m - mb);
        __syncthreads();
        if (tid == 0) nlist += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
      }
      const int n = nlist;
      for (int i = 0; i < n; ++i) {
        const float* row = dE + (mb + list[i]) * H + h0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (h0 + tid + 256 * q < H) acc[q] += row[tid + 256 * q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = h0 + tid + 256 * q;
      if (h < H) dW[(long)v * H + h] = accumulate ? dW[(long)v * H + h] + acc[q] : acc[q];
    }
  }
}

struct LstmWs {
  unsigned* flags;  // [16 words: status at word 0] [D*NC step flags], zeroed per launch
  size_t sync_bytes, nflags;
  float* hx;
  size_t hx_bytes;
  float* wp;   // (D*4H, I) permuted input weights; reused as dW_ih' in backward
  float* bp;   // (D*4H)
  float* dwhh; // (D*4H, H) scratch for dW_hh'
  unsigned long long* dbg;  // 256 workgroups x 8 phase counters (diagnostics)
  float* dbp;   // v4 backward: per-(group, row) time sums of dG (input side | hidden side), (2, D*G*NBR, 4H)
  size_t dbp_half;  // floats per side
  void* scratch;  // split-K slabs of the weight-gradient GEMMs / column-sum partials
  size_t scratch_bytes;
  // half-pair operand planes of the big products (gemm_hp.hip); null when the shape stays on gemm.hip
  bool hp;
  char* hp_x;    // fwd: x (T*B, I)            bwd: x^T (I, T*B)
  char* hp_w;    // fwd: W_ih' (D*4H, I)       bwd: W_ih'^T (I, D*4H)
  char* hp_dg;   // bwd: dG (T*B, D*4H)
  char* hp_dgt;  // bwd: dG^T (D*4H, T*B)
  char* hp_yt;   // bwd: time-shifted h^T per direction (D, H, T*B)
  uint32_t* hp_amax;  // per-row maxima of the operands: [M | N4 | M | N4 | I | I | D*H | N4] words (carve_lstm)
  size_t total;
};

// the hp path pays for the big products only (the operand conversion passes are fixed costs)
inline bool use_hp(int T, int B, int I, int H, int D) {
  if (getenv("RNNT_GEMM_NO_HP")) return false;
  const long M = (long)T * B, N4 = (long)D * 4 * H;
  // gemm_hp.hip addresses an operand's planes with 32-bit buffer offsets: a shape with a plane of 4 GB or more (e.g. bi-H = 1024,
  // B = 64, T = 2048) stays on gemm.hip.  The bound is evaluated with the widest input a layer of this stack can see (I or D*H),
  // so the sizing query (rnnt_hip_lstm_workspace_bytes) and every layer's launch take the same decision.
  const long Iw = I > D * H ? I : (long)D * H;
  const size_t lim = (size_t)1 << 32;
  if (hp_plane_bytes(M, Iw) >= lim || hp_plane_bytes(Iw, M) >= lim || hp_plane_bytes(M, N4) >= lim || hp_plane_bytes(N4, M) >= lim ||
      hp_plane_bytes(H, M) >= lim)
    return false;
  return M >= 1024 && N4 >= 512 && H >= 128 && (getenv("RNNT_GEMM_FORCE_HP") || (M * N4 >= (1l << 22)));
}

struct Plan {
  int Hs, NC, MT, NT, Bp, LDW, wgs_per_cu;
  size_t lds_fwd, lds_bwd;
};

bool make_plan(int B, int H, int D, int cus, Plan* pl) {
  if (H < 4 || H % 4 != 0 || B < 1 || B > 64 || D < 1 || D > 2) return false;
  const int NT = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
  pl->NT = NT;
  pl->Bp = 16 * NT;
  const int KG = (H + 15) / 16;
  pl->LDW = 16 * KG + 4;
  for (int per_cu = 1; per_cu <= 2; ++per_cu) {
    for (int Hs = 4; Hs <= 16; Hs *= 2) {
      if (H % Hs != 0) continue;
      const int MT = Hs / 4, NC = H / Hs;
      const size_t lds_fwd = (size_t)4 * Hs * pl->LDW * 4 + (size_t)4 * MT * NT * 64 * 16 + 16;
      const size_t lds_bwd = (size_t)KG * MT * 64 * 16 + (size_t)16 * MT * 4 * NT * 16 * 4 + 16;
      const size_t lds = lds_fwd > lds_bwd ? lds_fwd : lds_bwd;
      if (lds * per_cu > 160 * 1024) continue;
      if ((long)D * NC > (long)cus * per_cu) continue;
      pl->Hs = Hs; pl->NC = NC; pl->MT = MT; pl->wgs_per_cu = per_cu;
      pl->lds_fwd = lds_fwd; pl->lds_bwd = lds_bwd;
      return true;
    }
  }
  return false;
}


// v2 decomposition: largest hidden slice whose W_hh rows fit LDS (fewest workgroups per sync group), then as many
// batch slices as the CUs allow.  Returns false when the shape does not fit (caller falls back to v1).
bool make_plan2(int B, int H, int D, int cus, Plan2* pl) {
  if (H < 4 || H % 4 != 0 || B < 1 || D < 1 || D > 2) return false;
  if (getenv("RNNT_LSTM_V1")) return false;
  const int Kp = (int)align_up((size_t)H, 64);
  for (int HS = 16; HS >= 4; HS /= 2) {
    if (H % HS != 0) continue;
    const int NC = H / HS;
    const int Gmax = cus / (D * NC);
    if (Gmax < 1) continue;
    int G = (int)ceil_div(B, 4);
    if (G > Gmax) G = Gmax;
    const int Bg = (int)ceil_div(B, G);
    if (Bg > 16) continue;
    G = (int)ceil_div(B, Bg);
    const int BQ = Bg <= 4 ? 1 : (Bg <= 8 ? 2 : 4);
    const int KSf = 16 / HS, KSb = 64 / HS;
    const size_t wa = (size_t)16 * Kp * HS;
    const size_t part = (size_t)4 * BQ * 64 * 16;
    const size_t lds_f = wa + (size_t)4 * (1024 + 4 * BQ * KSf * 4) * 4 + part + 16;
    const size_t lds_b = wa + (size_t)4 * (1024 + 4 * BQ * KSb * 4) * 4 + part + 16;
    if (lds_f > 160 * 1024 || lds_b > 160 * 1024) continue;
    pl->HS = HS; pl->NC = NC; pl->G = G; pl->Bg = Bg; pl->BQ = BQ; pl->Kp = Kp;
    pl->lds_fwd = lds_f; pl->lds_bwd = lds_b;
    return true;
  }
  return false;
}

// v3 / v4 (W_hh in registers, bf16 pieces): 16 units per workgroup, H a multiple of 128 with H/128 among the instantiated
// k-step counts (H <= 640), no LDS constraint.
bool make_plan3(int B, int H, int D, int cus, bool bwd, Plan2* pl) {
  if (getenv("RNNT_LSTM_V1") || getenv("RNNT_LSTM_V2")) return false;
  if (H % 128 != 0 || B < 1 || D < 1 || D > 2) return false;
  const int nks = H / 128;
  // H = 768 / 1024: 8 waves x 3 / 4 k-steps (4 waves would need 288 / 384 operand registers per lane).
  // H = 640: 20 units per workgroup (5 blocks), 8 waves, K padded to 768 -> 32 workgroups per sync group, which fit one XCD
  //          (the 16-unit form has 40 and runs the write-through exchange: measured 4.9 / 7.1 us per step vs 3.2 / 2.9)
  if (!(nks >= 1 && nks <= 6) && nks != 8) return false;
  // (its backward keeps one of the six output blocks' W_hh pieces in LDS: with all 216 operand registers per lane it spilled 66-80
  //  and lost to the 16-unit / 4-wave form, 66.0 vs 63.8 ms per c5 step; with 180 it takes 45.5)
  (void)bwd;
  const bool h640 = nks == 5 && !getenv("RNNT_LSTM_NO_H640_FORM");
  if ((nks >= 6 || h640) && getenv("RNNT_LSTM_NO_8WAVE")) return false;
  const int MB = h640 ? 5 : 4, HS = 4 * MB;
  const int NC = H / HS;
  const int Gmax = cus / (D * NC);
  if (Gmax < 1) return false;
  int G = (int)ceil_div(B, 4);
  // (as many groups as the chip takes: a step's cost grows with the rows a group exchanges — 16-row instead of 8-row groups at c2:
  //  39.8 vs 32.0 ms per step)
  if (G > Gmax) G = Gmax;
  const int Bg = (int)ceil_div(B, G);
  if (Bg > 16) return false;
  G = (int)ceil_div(B, Bg);
  pl->HS = HS; pl->NC = NC; pl->G = G; pl->Bg = Bg; pl->BQ = Bg <= 4 ? 1 : (Bg <= 8 ? 2 : 4);
  pl->Kp = h640 ? 768 : H;
  pl->MB = MB;
  const int nwv = (nks >= 6 || h640) ? 8 : 4;
  const int ksb = (16 * MB + 31) / 32;
  pl->lds_fwd = (size_t)nwv * MB * 64 * 16 + 16;
  pl->lds_bwd = (size_t)nwv * 64 * 16 + 16 * (32 * ksb + 4) * 4 + 16;
  return true;
}

LstmWs carve_lstm(void* ws, int T, int B, int I, int H, int D, const Plan& pl) {
  LstmWs w;
  char* p = reinterpret_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += align_up(bytes, 256); return q; };
  size_t nflags = (size_t)D * pl.NC;
  size_t hxb = (size_t)2 * D * H * pl.Bp * 4 * 4;  // v1, sized for the backward exchange (B x 4H), fwd uses a quarter
  {
    Plan2 p2;
    int cus = device_cus();
    if (cus <= 0) cus = 256;
    if (make_plan2(B, H, D, cus, &p2)) {
      const size_t nf2 = (size_t)D * p2.G * p2.NC;
      const size_t hx2 = (size_t)2 * D * p2.G * 4 * p2.BQ * 4 * p2.Kp * 4;
      if (nf2 > nflags) nflags = nf2;
      if (hx2 > hxb) hxb = hx2;
    }
    for (int bwd = 0; bwd < 2; ++bwd) {
      if (!make_plan3(B, H, D, cus, bwd != 0, &p2)) continue;
      const size_t nf3 = (size_t)D * p2.G * p2.NC;
      const size_t hx3 = bwd ? (size_t)2 * D * p2.G * p2.NC * 4 * p2.BQ * p2.Kp * 4   // v4: per-producer partial dh, fp32
                             : (size_t)2 * D * p2.G * 4 * p2.BQ * p2.Kp * 6;          // v3: three bf16 planes of h
      if (nf3 > nflags) nflags = nf3;
      if (hx3 > hxb) hxb = hx3;
    }
  }
  w.nflags = nflags;
  w.sync_bytes = align_up((2 * nflags + 16) * 4, 16);  // status block | step flags | XCC table
  w.flags = reinterpret_cast<unsigned*>(take(w.sync_bytes));
  w.hx_bytes = hxb;
  w.hx = reinterpret_cast<float*>(take(w.hx_bytes));
  w.wp = reinterpret_cast<float*>(take((size_t)D * 4 * H * I * 4));
  w.bp = reinterpret_cast<float*>(take((size_t)D * 4 * H * 4));
  w.dwhh = reinterpret_cast<float*>(take((size_t)D * 4 * H * H * 4));
  w.dbg = reinterpret_cast<unsigned long long*>(take(512 * 8 * 8));
  w.dbp_half = (size_t)D * (B + 64) * 4 * H;  // D*G*NBR <= D*(B + 4*G) rows of 4H
  w.dbp = reinterpret_cast<float*>(take(2 * w.dbp_half * 4));
  {
    const int64_t M = (int64_t)T * B, N4 = (int64_t)D * 4 * H;
    size_t sc = rnnt_hip_gemm_workspace_bytes(N4, I, M);
    const size_t s2 = rnnt_hip_gemm_workspace_bytes(4 * H, H, M > B ? M - B : 1);
    const size_t s3 = rnnt_hip_colsum_workspace_bytes(M, N4);
    if (s2 > sc) sc = s2;
    if (s3 > sc) sc = s3;
    w.hp = use_hp(T, B, I, H, D);
    if (w.hp) {
      const size_t h1 = hp_gemm_workspace_bytes(N4, I, M), h2 = hp_gemm_workspace_bytes(4 * H, H, M);
      if (h1 > sc) sc = h1;
      if (h2 > sc) sc = h2;
      const int64_t mn[3] = {N4 * I, (int64_t)4 * H * H, (int64_t)4 * H * H};   // the grouped launch keeps all slabs at once
      const size_t h3 = HPQ_HEADER_BYTES + hp_gemm_grouped_workspace_bytes(mn, 1 + D);   // queue counters in front of the slabs
      if (h3 > sc) sc = h3;
    }
    w.scratch_bytes = sc;
    w.scratch = take(sc);
    w.hp_x = w.hp_w = w.hp_dg = w.hp_dgt = w.hp_yt = nullptr;
    w.hp_amax = nullptr;
    if (w.hp) {
      w.hp_amax = reinterpret_cast<uint32_t*>(take((size_t)(2 * M + 3 * N4 + 2 * I + (int64_t)D * H) * 4));
      w.hp_x = take(hp_plane_bytes(M, I) > hp_plane_bytes(I, M) ? hp_plane_bytes(M, I) : hp_plane_bytes(I, M));
      w.hp_w = take(hp_plane_bytes(N4, I) > hp_plane_bytes(I, N4) ? hp_plane_bytes(N4, I) : hp_plane_bytes(I, N4));
      w.hp_dg = take(hp_plane_bytes(M, N4));
      w.hp_dgt = take(hp_plane_bytes(N4, M));
      w.hp_yt = take((size_t)D * hp_plane_bytes(H, M));
    }
  }
  w.total = off;
  return w;
}

template <typename K>
int launch_persistent(K kernel, const LstmK& k, const Plan& pl, size_t lds, hipStream_t s, const char* what) {
  if (lds > 64 * 1024)
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  RNNT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds));
  const int cus = device_cus();
  if (per_cu < 1 || (long)k.D * k.NC > (long)cus * (per_cu < pl.wgs_per_cu ? per_cu : pl.wgs_per_cu)) {
    set_error("%s: %d workgroups cannot be co-resident (%d CUs x %d per CU)", what, k.D * k.NC, cus, per_cu);
    return RNNT_ERR_UNSUPPORTED;
  }
  {
    // algorithmic bytes of the recurrence: gates read+write (4H), c (H), y or dy (H) per (t,b,d) [+ weights once]
    const double per_tb = (k.dy ? (8.0 + 2.0 + 1.0) : (8.0 + 1.0 + 1.0)) * k.H * 4.0;
    ProfScope prof(k.dy ? RNNT_K_LSTM_BWD : RNNT_K_LSTM_FWD,
                   per_tb * k.T * k.B * k.D + 4.0 * 4.0 * k.H * k.H * k.D, s);
    hipLaunchKernelGGL(kernel, dim3(k.D * k.NC), dim3(256), lds, s, k);
  }
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

#define DISPATCH_HS_BQ_C(KERNEL, C, pl, ...)                                                \
  do {                                                                                      \
    const int key_ = (pl).HS * 10 + (pl).BQ;                                                \
    switch (key_) {                                                                         \
      case 41: rc = launch_persistent2(KERNEL<4, 1, C>, __VA_ARGS__); break;                \
      case 42: rc = launch_persistent2(KERNEL<4, 2, C>, __VA_ARGS__); break;                \
      case 44: rc = launch_persistent2(KERNEL<4, 4, C>, __VA_ARGS__); break;                \
      case 81: rc = launch_persistent2(KERNEL<8, 1, C>, __VA_ARGS__); break;                \
      case 82: rc = launch_persistent2(KERNEL<8, 2, C>, __VA_ARGS__); break;                \
      case 84: rc = launch_persistent2(KERNEL<8, 4, C>, __VA_ARGS__); break;                \
      case 161: rc = launch_persistent2(KERNEL<16, 1, C>, __VA_ARGS__); break;              \
      case 162: rc = launch_persistent2(KERNEL<16, 2, C>, __VA_ARGS__); break;              \
      case 164: rc = launch_persistent2(KERNEL<16, 4, C>, __VA_ARGS__); break;              \
      default: set_error("lstm: no v2 kernel for HS=%d BQ=%d", (pl).HS, (pl).BQ); rc = RNNT_ERR_UNSUPPORTED; \
    }                                                                                       \
  } while (0)
#define DISPATCH_HS_BQ(KERNEL, cell, pl, ...)                                               \
  do {                                                                                      \
    if ((cell) == RNNT_CELL_LSTM) DISPATCH_HS_BQ_C(KERNEL, 0, pl, __VA_ARGS__);             \
    else if ((cell) == RNNT_CELL_GRU) DISPATCH_HS_BQ_C(KERNEL, 1, pl, __VA_ARGS__);         \
    else DISPATCH_HS_BQ_C(KERNEL, 2, pl, __VA_ARGS__);                                      \
  } while (0)

int check_desc(const rnnt_lstm_desc* d, Plan* pl, LstmWs* w) {
  RNNT_CHECK_ARG(d != nullptr, "lstm: null descriptor");
  RNNT_CHECK_ARG(d->T >= 1 && d->I >= 1, "lstm: T and I must be positive (T=%d I=%d)", d->T, d->I);
  const int cus = device_cus();
  RNNT_CHECK_ARG(cus > 0, "lstm: no HIP device");
  if (!make_plan(d->B, d->H, d->D, cus, pl)) {
    set_error("lstm: unsupported configuration B=%d H=%d D=%d (need H%%4==0, 1<=B<=64, D in {1,2}, slice must fit %d CUs)",
              d->B, d->H, d->D, cus);
    return RNNT_ERR_UNSUPPORTED;
  }
  RNNT_CHECK_ARG(d->cell >= RNNT_CELL_LSTM && d->cell <= RNNT_CELL_RNN_RELU, "lstm: unknown cell type %d", d->cell);
  RNNT_CHECK_ARG(d->lens && d->x && d->y && d->gates && (d->cst || d->cell != RNNT_CELL_LSTM), "lstm: null tensor");
  for (int k = 0; k < d->D; ++k)
    RNNT_CHECK_ARG(d->w_ih[k] && d->w_hh[k] && d->b_ih[k] && d->b_hh[k], "lstm: null weight (direction %d)", k);
  RNNT_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "lstm: dropout_p must be in [0,1)");
  RNNT_CHECK_ARG(d->dropout_p == 0.f || d->y_drop, "lstm: dropout_p > 0 needs y_drop");
  RNNT_CHECK_ARG(d->x_abs_bound >= 0.f && d->x_abs_bound < 1e30f, "lstm: x_abs_bound must be 0 (measure) or a finite positive bound");
  RNNT_CHECK_ARG(!d->row_idx || (d->n_rows >= 1 && d->n_rows <= (int64_t)d->T * d->B), "lstm: row_idx needs 1 <= n_rows <= T*B (got %d)", d->n_rows);
  *w = carve_lstm(d->workspace, d->T, d->B, d->I, d->H, d->D, *pl);
  RNNT_CHECK_ARG(d->workspace && d->workspace_bytes >= w->total, "lstm: workspace too small (%zu < %zu)",
                 d->workspace_bytes, w->total);
  RNNT_CHECK_ARG((reinterpret_cast<uintptr_t>(d->gates) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(d->cst) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->aux) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(d->workspace) & 255) == 0,
                 "lstm: gates/y/cst must be 16-byte aligned, workspace 256-byte aligned");
  return RNNT_OK;
}

void fill_kernel_args(const rnnt_lstm_desc* d, const Plan& pl, const LstmWs& w, LstmK* k) {
  k->T = d->T; k->B = d->B; k->H = d->H; k->D = d->D;
  k->Hs = pl.Hs; k->NC = pl.NC; k->Bp = pl.Bp; k->LDW = pl.LDW;
  k->lens = d->lens; k->gates = d->gates; k->cst = d->cst; k->y = d->y;
  k->ydrop = d->dropout_p > 0.f ? d->y_drop : nullptr;
  k->keep_scale = d->dropout_p > 0.f ? 1.f / (1.f - d->dropout_p) : 1.f;
  k->drop_thresh = (unsigned)((double)d->dropout_p * 4294967296.0);
  k->seed = d->dropout_seed;
  k->w_hh[0] = d->w_hh[0]; k->w_hh[1] = d->D > 1 ? d->w_hh[1] : d->w_hh[0];
  // status word: the caller's sticky device word when given (never reset by the library: a raised status makes every later
  // launch bail out at its first wait and stays visible until the caller reads it), else word 0 of the workspace (reset per launch)
  k->hx = w.hx; k->status = d->status ? d->status : w.flags; k->flags = w.flags + 16;
  k->dy = nullptr;
  k->cell = d->cell;
  k->b_hh[0] = d->b_hh[0]; k->b_hh[1] = d->D > 1 ? d->b_hh[1] : d->b_hh[0];
  k->aux = d->aux;
  k->G = 1; k->Bg = d->B; k->Kp = d->H; k->NGL = d->D;
  k->dbp = w.dbp; k->dbhp = w.dbp + w.dbp_half;
  k->dbg = getenv("RNNT_LSTM_DBG") ? w.dbg : nullptr;
  k->xcc = w.flags + 16 + w.nflags;
  k->allow_local = getenv("RNNT_LSTM_NO_XCD_LOCAL") ? 0 : 1;
  k->hw_math = getenv("RNNT_LSTM_EXACT_MATH") ? 0 : 1;
  k->pause = 0;
  k->gbound = 0;
  k->colmax = k->colmax_h = nullptr;
  k->rowmax = nullptr;
}

// rnnt_lstm_desc.row_idx is honoured where EVERY consumer of the stash gathers the valid rows: v5 recurrences (group step bounds) with
// the half-pair products (row gather in the operand fetch / k-gather in the transposed splits).  Same answer in the forward and the
// backward call of a layer (it depends on the shape only).
bool shape_takes_row_idx(int T, int B, int I, int H, int D, int cell, int cus) {
  Plan2 p2;
  return use_hp(T, B, I, H, D) && I >= 32 && T > 1 && make_plan3(B, H, D, cus, true, &p2) && lstm5_supported(T, B, H, D, cell);
}
bool ragged_plan(const rnnt_lstm_desc* d, const LstmWs& w, int cus) {
  return d->row_idx && d->n_rows > 0 && d->n_rows < (int64_t)d->T * d->B && w.hp && d->x_sb == d->I && d->x_st == (int64_t)d->B * d->I &&
         shape_takes_row_idx(d->T, d->B, d->I, d->H, d->D, d->cell, cus);
}

}  // namespace
}  // namespace rnnt

using namespace rnnt;

extern "C" int32_t rnnt_hip_lstm_max_batch(int32_t H, int32_t D, int32_t cell) {
  // largest per-call batch the persistent kernels take for this shape (callers split bigger batches along B:
  // sequences are independent, weight gradients add)
  int cus = device_cus();
  if (cus <= 0) cus = 256;
  int best = 0;
  for (int B = 64; B >= 1; --B) {
    Plan2 p2;
    Plan p1;
    // the backward decides: a batch both directions of the recurrence can take
    if (make_plan3(B, H, D, cus, true, &p2) || make_plan2(B, H, D, cus, &p2) || (cell == RNNT_CELL_LSTM && make_plan(B, H, D, cus, &p1))) { best = B; break; }
  }
  return best;
}

// the placement rule shared by the backward (which XCDs its grouped products avoid) and the caller's decision to overlap at all
static int recurrence_xcds(int T, int B, int H, int D, int cell, int cus) {
  Plan2 p2;
  if (!make_plan3(B, H, D, cus, true, &p2) || !lstm5_supported(T, B, H, D, cell)) return 8;
  const int NG = D * p2.G;
  return (NG <= 4 && p2.NC <= 32 && !getenv("RNNT_LSTM_NO_XCD_STRIDE")) ? NG : 8;   // launch_persistent2: stride 8, group g on XCD g
}

extern "C" int32_t rnnt_hip_lstm_takes_row_idx(int32_t T, int32_t B, int32_t I, int32_t H, int32_t D, int32_t cell) {
  int cus = device_cus();
  if (cus <= 0) cus = 256;
  if (T < 1 || B < 1 || I < 1 || H < 4 || D < 1 || D > 2) return 0;
  return shape_takes_row_idx(T, B, I, H, D, cell, cus) ? 1 : 0;
}

extern "C" int32_t rnnt_hip_lstm_free_xcds(int32_t T, int32_t B, int32_t H, int32_t D, int32_t cell) {
  int cus = device_cus();
  if (cus <= 0) cus = 256;
  if (T < 1 || B < 1 || H < 4 || D < 1 || D > 2) return 0;
  return 8 - recurrence_xcds(T, B, H, D, cell, cus);
}

extern "C" size_t rnnt_hip_lstm_workspace_bytes(int32_t T, int32_t B, int32_t I, int32_t H, int32_t D) {
  Plan pl;
  int cus = device_cus();
  if (cus <= 0) cus = 256;  // sizing query without a device: assume MI355X
  if (T < 1 || I < 1 || !make_plan(B, H, D, cus, &pl)) return 0;
  return carve_lstm(nullptr, T, B, I, H, D, pl).total;
}

#define DISPATCH_MT_NT(KERNEL, pl, ...)                                                     \
  do {                                                                                      \
    const int key_ = (pl).MT * 10 + (pl).NT;                                                \
    switch (key_) {                                                                         \
      case 11: rc = launch_persistent(KERNEL<1, 1>, __VA_ARGS__); break;                    \
      case 12: rc = launch_persistent(KERNEL<1, 2>, __VA_ARGS__); break;                    \
      case 14: rc = launch_persistent(KERNEL<1, 4>, __VA_ARGS__); break;                    \
      case 21: rc = launch_persistent(KERNEL<2, 1>, __VA_ARGS__); break;                    \
      case 22: rc = launch_persistent(KERNEL<2, 2>, __VA_ARGS__); break;                    \
      case 24: rc = launch_persistent(KERNEL<2, 4>, __VA_ARGS__); break;                    \
      case 41: rc = launch_persistent(KERNEL<4, 1>, __VA_ARGS__); break;                    \
      case 42: rc = launch_persistent(KERNEL<4, 2>, __VA_ARGS__); break;                    \
      case 44: rc = launch_persistent(KERNEL<4, 4>, __VA_ARGS__); break;                    \
      default: set_error("lstm: no kernel for MT=%d NT=%d", (pl).MT, (pl).NT); rc = RNNT_ERR_UNSUPPORTED; \
    }                                                                                       \
  } while (0)

extern "C" int rnnt_hip_lstm_fwd(const rnnt_lstm_desc* d, void* stream) {
  Plan pl;
  LstmWs w;
  if (int rc = check_desc(d, &pl, &w)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int H = d->H, D = d->D, I = d->I;
  const int ngate = d->cell == RNNT_CELL_LSTM ? 4 : (d->cell == RNNT_CELL_GRU ? 3 : 1);
  // 1. gate-adjacent copy of W_ih (both directions stacked) and of b_ih + b_hh
  {
    const long per = (long)4 * H * I;
    hipLaunchKernelGGL(permute_w_kernel, dim3((unsigned)ceil_div(per, 256), D), dim3(256), 0, s, d->w_ih[0],
                       D > 1 ? d->w_ih[1] : d->w_ih[0], H, I, ngate, w.wp);
    RNNT_CHECK_LAUNCH();
    hipLaunchKernelGGL(permute_bias_kernel, dim3((unsigned)ceil_div(4 * H, 256), D), dim3(256), 0, s, d->b_ih[0], d->b_hh[0],
                       D > 1 ? d->b_ih[1] : d->b_ih[0], D > 1 ? d->b_hh[1] : d->b_hh[0], H, ngate,
                       d->cell == RNNT_CELL_GRU ? 1 : 0, w.bp);
    RNNT_CHECK_LAUNCH();
  }
  // 2. hoisted input projection for all timesteps: gates[(t,b)][d*4H + 4j+g] = x(t,b,:) . W_ih'[.] + bias'
  const bool x_plain = d->x_sb == I && d->x_st == (int64_t)d->B * I;
  const bool ragged = ragged_plan(d, w, device_cus());   // valid frames only (rnnt_lstm_desc.row_idx)
  if (w.hp && x_plain && I >= 32) {  // f16 matrix cores on half-pair operands (gemm_hp.hip)
    const int64_t M = (int64_t)d->T * d->B, N4 = (int64_t)D * 4 * H;
    const int64_t Mv = ragged ? d->n_rows : M;             // rows the product runs over
    const int* ridx = ragged ? d->row_idx : nullptr;
    uint32_t* ax = w.hp_amax, *aw = w.hp_amax + M;
    if (int rc = hp_split(d->x, Mv, I, I, ax, w.hp_x, s, ridx)) return rc;   // planes / maxima of the valid rows, in place
    if (int rc = hp_split(w.wp, N4, I, I, aw, w.hp_w, s)) return rc;
    if (int rc = hp_gemm(w.hp_x, ax, w.hp_w, aw, Mv, N4, I, d->gates, 1, N4, 0, w.bp, 0, nullptr, 0, s, ridx, M, ridx)) return rc;
  } else {
    rnnt_gemm_desc g = {};
    g.M = (int64_t)d->T * d->B; g.N = (int64_t)D * 4 * H; g.K = I;
    g.A = d->x; g.a_div = d->B; g.a_so = d->x_st; g.a_si = d->x_sb; g.a_sk = 1; g.a_mc = 0;
    g.B = w.wp; g.b_sn = I; g.b_sk = 1;
    g.C = d->gates; g.c_div = 1; g.c_so = g.N; g.c_si = 0;
    g.bias = w.bp;
    if (int rc = rnnt_hip_gemm_f32(&g, s)) return rc;
  }
  // 3. the recurrence
  // sync block and exchange buffers are neighbours in the workspace (carve_lstm): one fill
  RNNT_CHECK_HIP(hipMemsetAsync(w.flags, 0, (size_t)(reinterpret_cast<char*>(w.hx) - reinterpret_cast<char*>(w.flags)) + w.hx_bytes, s));
  LstmK k;
  fill_kernel_args(d, pl, w, &k);
  k.gbound = ragged ? 1 : 0;
  int rc = RNNT_OK;
  Plan2 p2;
  const int cus = device_cus();
  auto adopt = [&](const Plan2& q) {
    k.NC = q.NC; k.Hs = q.HS; k.G = q.G; k.Bg = q.Bg; k.Kp = q.Kp;
  };
  if (make_plan3(d->B, d->H, d->D, cus, false, &p2)) {
    adopt(p2);
    const int nks = p2.Kp / 128;
#define LAUNCH_V3(N)                                                                                              \
    do {                                                                                                          \
      if (d->cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd3_kernel<N, 0>, k, p2, p2.lds_fwd, s, "lstm_fwd3"); \
      else if (d->cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd3_kernel<N, 1>, k, p2, p2.lds_fwd, s, "lstm_fwd3"); \
      else rc = launch_persistent2(lstm_fwd3_kernel<N, 2>, k, p2, p2.lds_fwd, s, "lstm_fwd3");                     \
    } while (0)
    if (lstm5_supported(d->T, d->B, d->H, d->D, d->cell)) {  // v5: tagged-payload exchange, f16 matrix cores (lstm5.hip)
      rc = lstm5_fwd_launch(k, p2, d->cell, s);
    } else if (p2.MB == 5) {  // H = 640: 5 blocks, 8 waves x 3 k-steps over K padded to 768
      const size_t lds5 = p2.lds_fwd + 8 * 1 * 3 * 3 * 1024;  // one of the five blocks' pieces in LDS
      if (d->cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd3_kernel<3, 0, 8, 5, 1>, k, p2, lds5, s, "lstm_fwd3", 512);
      else if (d->cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd3_kernel<3, 1, 8, 5, 1>, k, p2, lds5, s, "lstm_fwd3", 512);
      else rc = launch_persistent2(lstm_fwd3_kernel<3, 2, 8, 5, 1>, k, p2, lds5, s, "lstm_fwd3", 512);
    } else if (nks == 4 && !getenv("RNNT_LSTM_NO_8WAVE")) {  // H = 512 forward: 8 waves x 2 k-steps (12.5 vs 13.1 ms per c2 step)
      p2.lds_fwd = (size_t)8 * 4 * 64 * 16 + 16;
      if (d->cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd3_kernel<2, 0, 8>, k, p2, p2.lds_fwd, s, "lstm_fwd3", 512);
      else if (d->cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd3_kernel<2, 1, 8>, k, p2, p2.lds_fwd, s, "lstm_fwd3", 512);
      else rc = launch_persistent2(lstm_fwd3_kernel<2, 2, 8>, k, p2, p2.lds_fwd, s, "lstm_fwd3", 512);
    } else if (nks == 1) LAUNCH_V3(1);
    else if (nks == 2) LAUNCH_V3(2);
    else if (nks == 3) LAUNCH_V3(3);
    else if (nks == 4) LAUNCH_V3(4);
    else if (nks == 5) LAUNCH_V3(5);
    else {  // H = 768 / 1024: 8 waves x 3 / 4 k-steps
#define LAUNCH_V38(N)                                                                                                       \
      do {  /* one of the four blocks' pieces in LDS */                                                                     \
        const size_t lds8 = p2.lds_fwd + 8 * 1 * (N) * 3 * 1024;                                                            \
        if (d->cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd3_kernel<N, 0, 8, 4, 1>, k, p2, lds8, s, "lstm_fwd3", 512); \
        else if (d->cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd3_kernel<N, 1, 8, 4, 1>, k, p2, lds8, s, "lstm_fwd3", 512); \
        else rc = launch_persistent2(lstm_fwd3_kernel<N, 2, 8, 4, 1>, k, p2, lds8, s, "lstm_fwd3", 512);                     \
      } while (0)
      if (nks == 6) LAUNCH_V38(3);
      else LAUNCH_V38(4);
#undef LAUNCH_V38
    }
#undef LAUNCH_V3
  } else if (make_plan2(d->B, d->H, d->D, cus, &p2)) {
    adopt(p2);
    DISPATCH_HS_BQ(lstm_fwd2_kernel, d->cell, p2, k, p2, p2.lds_fwd, s, "lstm_fwd2");
  } else if (d->cell == RNNT_CELL_LSTM) {
    DISPATCH_MT_NT(lstm_fwd_kernel, pl, k, pl, pl.lds_fwd, s, "lstm_fwd");
  } else {
    set_error("rnn: GRU / Elman cells need the grouped decomposition (B/G <= 16 rows per group); B=%d H=%d does not fit", d->B, d->H);
    rc = RNNT_ERR_UNSUPPORTED;
  }
  return rc;
}

extern "C" int rnnt_hip_lstm_bwd(const rnnt_lstm_bwd_desc* bd, void* stream) {
  RNNT_CHECK_ARG(bd != nullptr, "lstm_bwd: null descriptor");
  const rnnt_lstm_desc* d = &bd->f;
  Plan pl;
  LstmWs w;
  if (int rc = check_desc(d, &pl, &w)) return rc;
  RNNT_CHECK_ARG(bd->dy, "lstm_bwd: null dy");
  const int T = d->T, B = d->B, H = d->H, D = d->D, I = d->I;
  const bool gru = d->cell == RNNT_CELL_GRU;
  const int ngate = d->cell == RNNT_CELL_LSTM ? 4 : (gru ? 3 : 1);
  RNNT_CHECK_ARG(!gru || d->aux, "lstm_bwd: GRU needs the aux buffer (T,B,D*4H)");
  for (int k = 0; k < D; ++k) RNNT_CHECK_ARG(!gru || bd->db_hh[k], "lstm_bwd: GRU needs db_hh");
  const int acc = bd->accumulate ? 1 : 0;       // += into dw_ih / dw_hh / db / db_hh (flat-gradient views) instead of =
  // LSTM / Elman: grad b_hh == grad b_ih; when the caller also hands db_hh it receives the same values (second destination)
  float* twin0 = gru ? nullptr : bd->db_hh[0];
  float* twin1 = gru ? nullptr : (D > 1 ? bd->db_hh[1] : bd->db_hh[0]);
  if (twin0 == bd->db[0]) twin0 = nullptr;
  if (twin1 == (D > 1 ? bd->db[1] : bd->db[0])) twin1 = nullptr;
  const float* ghid = gru ? d->aux : d->gates;  // hidden-side gate gradients (== input side except for GRU's n gate)
  RNNT_CHECK_ARG(d->x_sb == I && d->x_st == (int64_t)B * I, "lstm_bwd: x must be time-major contiguous (T,B,I)");
  for (int k = 0; k < D; ++k) RNNT_CHECK_ARG(bd->dw_ih[k] && bd->dw_hh[k] && bd->db[k], "lstm_bwd: null gradient output");
  hipStream_t s = (hipStream_t)stream;
  RNNT_CHECK_ARG(bd->phase >= RNNT_LSTM_BWD_ALL && bd->phase <= RNNT_LSTM_BWD_WEIGHTS, "lstm_bwd: phase must be 0, 1 or 2");
  // phase 1 = steps 1-2 (the chain autograd waits for), phase 2 = steps 3-5 (weight / bias gradients: any stream ordered after phase 1)
  const bool do_recur = bd->phase != RNNT_LSTM_BWD_WEIGHTS, do_weights = bd->phase != RNNT_LSTM_BWD_RECUR;

  // 1. reverse-time recurrence: gates (activated) -> dG in place
  if (do_recur)   // sync block and exchange buffers are neighbours in the workspace (carve_lstm): one fill
    RNNT_CHECK_HIP(hipMemsetAsync(w.flags, 0, (size_t)(reinterpret_cast<char*>(w.hx) - reinterpret_cast<char*>(w.flags)) + w.hx_bytes, s));
  LstmK k;
  fill_kernel_args(d, pl, w, &k);
  k.dy = bd->dy;
  int rc = RNNT_OK;
  Plan2 p2;
  const int cus = device_cus();
  const bool ragged = ragged_plan(d, w, cus);            // valid frames only (rnnt_lstm_desc.row_idx): as in the forward call
  const int* ridx = ragged ? d->row_idx : nullptr;
  k.gbound = ragged ? 1 : 0;
  auto adopt = [&](const Plan2& q) {
    k.NC = q.NC; k.Hs = q.HS; k.G = q.G; k.Bg = q.Bg; k.Kp = q.Kp;
  };
  bool fused_db = false, colmax_done = false, rowmax_done = false;
  int db_rows = 0;
  unsigned xcd_skip = 0;
  if (make_plan3(d->B, d->H, d->D, cus, true, &p2)) {
    adopt(p2);
    fused_db = true;
    db_rows = p2.G * 4 * p2.BQ;
    const int nks = p2.Kp / 128;
#define LAUNCH_V4_C(N, BQ_, C) rc = launch_persistent2(lstm_bwd4_kernel<2 * (N), BQ_, C>, k, p2, p2.lds_bwd, s, "lstm_bwd4")
#define LAUNCH_V4_B(N, C)                           \
    do {                                            \
      if (p2.BQ == 1) LAUNCH_V4_C(N, 1, C);         \
      else if (p2.BQ == 2) LAUNCH_V4_C(N, 2, C);    \
      else LAUNCH_V4_C(N, 4, C);                    \
    } while (0)
#define LAUNCH_V4(N)                                               \
    do {                                                           \
      if (d->cell == RNNT_CELL_LSTM) LAUNCH_V4_B(N, 0);            \
      else if (d->cell == RNNT_CELL_GRU) LAUNCH_V4_B(N, 1);        \
      else LAUNCH_V4_B(N, 2);                                      \
    } while (0)
    if (lstm5_supported(d->T, d->B, d->H, d->D, d->cell)) {  // v5 (lstm5.hip)
      if (w.hp) {  // the recurrence also leaves the column maxima of dG (the scales of the half-pair dG^T planes): no extra pass
        const int64_t Mr = (int64_t)T * B, N4r = (int64_t)D * 4 * H;
        k.colmax = w.hp_amax + Mr + N4r + Mr;
        k.colmax_h = w.hp_amax + 2 * Mr + 2 * N4r + 2 * I + (int64_t)D * H;
        // LSTM / Elman layers that hand on dx: the recurrence also leaves the row maxima, and ONE pass over dG then writes both
        // orientations of its planes (hp_split_both) instead of a row-major pass that measures each row first plus a transposed pass
        unsigned* fill_from = k.colmax;
        if (!gru && I >= 128 && bd->dx && !getenv("RNNT_GEMM_HP_NO_FUSED_SPLIT")) {
          k.rowmax = w.hp_amax + Mr + N4r;   // = a_dgr below, directly in front of the column table
          fill_from = k.rowmax;
          rowmax_done = true;
        }
        if (do_recur)   // one fill from the first table to the end of the last (the tables in between are written later)
          RNNT_CHECK_HIP(hipMemsetAsync(fill_from, 0, (size_t)(k.colmax_h + N4r - fill_from) * 4, s));
        colmax_done = true;
      }
      if (do_recur) rc = lstm5_bwd_launch(k, p2, d->cell, s);
      // the weight-gradient products of THIS layer may run beside the recurrence of the next one (same shape): that one sits on
      // XCDs 0 .. D*G-1 (launch_persistent2's stride-8 placement), the products keep to the others
      if (bd->beside_recurrence) {
        // (only on a device that exposes all 8 XCDs — the 256-CU SPX mode the placement rule was measured on; a partitioned device
        //  runs the products on every XCD it has, and gemm_hp.hip's check kernel raises the status word if a launch left units undone)
        const int used = recurrence_xcds(T, B, H, D, d->cell, cus);
        if (used <= 4 && cus == 256) xcd_skip = (1u << used) - 1u;
      }
    } else if (!do_recur) {
      // phase 2: nothing to launch here
    } else if (p2.MB == 5) {  // H = 640: own 80 gate columns (3 k-steps), 48 output blocks over 8 waves
#define LAUNCH_V45_C(BQ_, C) rc = launch_persistent2(lstm_bwd4_kernel<6, BQ_, C, 8, 5, 1>, k, p2, p2.lds_bwd + 8 * 1 * 3 * 3 * 1024, s, "lstm_bwd4", 512)
#define LAUNCH_V45_B(C)                           \
      do {                                        \
        if (p2.BQ == 1) LAUNCH_V45_C(1, C);       \
        else if (p2.BQ == 2) LAUNCH_V45_C(2, C);  \
        else LAUNCH_V45_C(4, C);                  \
      } while (0)
      if (d->cell == RNNT_CELL_LSTM) LAUNCH_V45_B(0);
      else if (d->cell == RNNT_CELL_GRU) LAUNCH_V45_B(1);
      else LAUNCH_V45_B(2);
#undef LAUNCH_V45_B
#undef LAUNCH_V45_C
    } else if (nks == 1) LAUNCH_V4(1);  // (H = 512 backward with 8 waves: 12.9 vs 11.9 ms per c2 step -> stays at 4)
    else if (nks == 2) LAUNCH_V4(2);
    else if (nks == 3) LAUNCH_V4(3);
    else if (nks == 4) LAUNCH_V4(4);
    else if (nks == 5) LAUNCH_V4(5);
    else {  // H = 768 / 1024: 8 waves, 6 / 8 output blocks each
#define LAUNCH_V48_C(N, BQ_, C) rc = launch_persistent2(lstm_bwd4_kernel<N, BQ_, C, 8, 4, 2>, k, p2, p2.lds_bwd + 8 * 2 * 2 * 3 * 1024, s, "lstm_bwd4", 512)
#define LAUNCH_V48_B(N, C)                           \
      do {                                           \
        if (p2.BQ == 1) LAUNCH_V48_C(N, 1, C);       \
        else if (p2.BQ == 2) LAUNCH_V48_C(N, 2, C);  \
        else LAUNCH_V48_C(N, 4, C);                  \
      } while (0)
#define LAUNCH_V48(N)                                          \
      do {                                                     \
        if (d->cell == RNNT_CELL_LSTM) LAUNCH_V48_B(N, 0);     \
        else if (d->cell == RNNT_CELL_GRU) LAUNCH_V48_B(N, 1); \
        else LAUNCH_V48_B(N, 2);                               \
      } while (0)
      if (nks == 6) LAUNCH_V48(6);
      else LAUNCH_V48(8);
#undef LAUNCH_V48
#undef LAUNCH_V48_B
#undef LAUNCH_V48_C
    }
#undef LAUNCH_V4
#undef LAUNCH_V4_B
#undef LAUNCH_V4_C
  } else if (make_plan2(d->B, d->H, d->D, cus, &p2)) {
    adopt(p2);
    if (do_recur) DISPATCH_HS_BQ(lstm_bwd2_kernel, d->cell, p2, k, p2, p2.lds_bwd, s, "lstm_bwd2");
  } else if (d->cell == RNNT_CELL_LSTM) {
    if (do_recur) DISPATCH_MT_NT(lstm_bwd_kernel, pl, k, pl, pl.lds_bwd, s, "lstm_bwd");
  } else {
    set_error("rnn: GRU / Elman cells need the grouped decomposition; B=%d H=%d does not fit", d->B, d->H);
    rc = RNNT_ERR_UNSUPPORTED;
  }
  if (rc) return rc;

  const int64_t M = (int64_t)T * B, N4 = (int64_t)D * 4 * H;
  const int64_t Mv = ragged ? d->n_rows : M;   // rows (row-major operands: gathered by the GEMM) / contraction length (transposed planes: packed)
  const bool hp_in = w.hp && I >= 128;   // products with I as an output / contraction width on the f16 matrix cores
  // per-row maxima: dG rows | dG columns | W_ih'^T rows | X^T rows | h^T rows
  uint32_t* a_dgr = w.hp ? w.hp_amax + M + N4 : nullptr;
  uint32_t* a_dgc = w.hp ? a_dgr + M : nullptr;
  uint32_t* a_w = w.hp ? a_dgc + N4 : nullptr;
  uint32_t* a_x = w.hp ? a_w + I : nullptr;
  uint32_t* a_y = w.hp ? a_x + I : nullptr;
  if (w.hp) {  // half-pair planes of dG in both orientations (gemm_hp.hip is NT-only: transposed operands are materialised)
    const bool both = rowmax_done && colmax_done && hp_in && bd->dx;   // (same in both phases of a two-phase backward)
    if (do_recur && hp_in && bd->dx) {
      if (both) rc = hp_split_both(d->gates, Mv, N4, N4, a_dgr, a_dgc, w.hp_dg, w.hp_dgt, s, ridx);
      else rc = hp_split(d->gates, Mv, N4, N4, a_dgr, w.hp_dg, s, ridx);
      if (rc) return rc;
    }
    if (do_weights && !colmax_done)
      if ((rc = hp_colmax(d->gates, M, N4, N4, a_dgc, s))) return rc;
    if (do_weights && !both)
      if ((rc = hp_split_t(d->gates, N4, Mv, N4, M, 0, a_dgc, w.hp_dgt, s, ridx))) return rc;
  }
  // 2. dX = dG . W_ih'   (needs the permuted weights: rebuild them, the forward copy may have been overwritten)
  if (do_recur && bd->dx) {
    const long per = (long)4 * H * I;
    hipLaunchKernelGGL(permute_w_kernel, dim3((unsigned)ceil_div(per, 256), D), dim3(256), 0, s, d->w_ih[0],
                       D > 1 ? d->w_ih[1] : d->w_ih[0], H, I, ngate, w.wp);
    RNNT_CHECK_LAUNCH();
    if (hp_in) {
      if ((rc = hp_colmax(w.wp, N4, I, I, a_w, s))) return rc;
      if ((rc = hp_split_t(w.wp, I, N4, I, N4, 0, a_w, w.hp_w, s))) return rc;   // W_ih'^T: (I, contraction N4)
      if ((rc = hp_gemm(w.hp_dg, a_dgr, w.hp_w, a_w, Mv, I, N4, bd->dx, 1, I, 0, nullptr, 0, nullptr, 0, s, ridx, M, ridx))) return rc;
    } else {
      rnnt_gemm_desc g = {};
      g.M = M; g.N = I; g.K = N4;
      g.A = d->gates; g.a_div = 1; g.a_so = N4; g.a_si = 0; g.a_sk = 1; g.a_mc = 0;
      g.B = w.wp; g.b_sn = 1; g.b_sk = I;
      g.C = bd->dx; g.c_div = 1; g.c_so = I; g.c_si = 0;
      if ((rc = rnnt_hip_gemm_f32(&g, s))) return rc;
    }
  }
  if (!do_weights) return RNNT_OK;
  // All three weight-gradient products on the half-pair path (LSTM / Elman) beside a recurrence: their operand planes first, then
  // ONE queue-driven launch (gemm_hp.hip) that stays off the recurrence's XCDs.
  // (alone on the device three launches are faster: 1.13 vs 1.29 ms for a c2 layer, tools/gemm_hpq_bench.py — the queue form is for
  // the overlapped case, where it keeps off the recurrence's XCDs)
  const bool grouped = w.hp && hp_in && T > 1 && !gru && D <= 2 && (xcd_skip != 0u || getenv("RNNT_GEMM_HP_GROUP")) && !getenv("RNNT_GEMM_HP_NO_GROUP");
  // 3. dW_ih' = dG^T . X  (both directions at once), un-permute rows into torch layout
  {
    // (a narrow input — the 80 mel bins of layer 0 — still goes to the half-pair kernel: one 256-wide tile column, split over K)
    const bool x_plain = d->x_sb == I && d->x_st == (int64_t)B * I;
    if (hp_in || (w.hp && I >= 32 && x_plain && !getenv("RNNT_GEMM_HP_NO_NARROW"))) {
      if (d->x_abs_bound > 0.f) {   // bounded input (the dropped output of the layer below): its bound is the scale, no pass over x
        uint32_t bits;
        memcpy(&bits, &d->x_abs_bound, 4);
        RNNT_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)a_x, (int)bits, (size_t)I, s));
      } else if ((rc = hp_colmax(d->x, M, I, I, a_x, s))) return rc;
      if ((rc = hp_split_t(d->x, I, Mv, I, M, 0, a_x, w.hp_x, s, ridx))) return rc;     // X^T: (I, contraction over the (valid) frames)
      if (!grouped)
        if ((rc = hp_gemm(w.hp_dgt, a_dgc, w.hp_x, a_x, N4, I, Mv, w.wp, 1, I, 0, nullptr, 0, w.scratch, w.scratch_bytes, s))) return rc;
    } else {
      rnnt_gemm_desc g = {};
      g.M = N4; g.N = I; g.K = M;
      g.A = d->gates; g.a_mc = 1; g.a_sk = N4; g.a_div = 1;
      g.B = d->x; g.b_sn = 1; g.b_sk = I;
      g.C = w.wp; g.c_div = 1; g.c_so = I; g.c_si = 0;
      g.workspace = w.scratch; g.workspace_bytes = w.scratch_bytes;
      if ((rc = rnnt_hip_gemm_f32(&g, s))) return rc;
    }
    if (!grouped) {
      const long per = (long)4 * H * I;
      hipLaunchKernelGGL(unpermute_w_kernel, dim3((unsigned)ceil_div(per, 256), D), dim3(256), 0, s, w.wp, H, I, per, ngate,
                         bd->dw_ih[0], D > 1 ? bd->dw_ih[1] : bd->dw_ih[0], acc);
      RNNT_CHECK_LAUNCH();
    }
  }
  // 4. dW_hh'[d] = sum_t dG[t]^T . h_prev(t): time-shifted views of dG and y (padded frames are zero in both)
  if (w.hp && T > 1) {
    if (d->cell != RNNT_CELL_RNN_RELU) {   // |h| < 1 for LSTM / GRU / tanh cells: the planes of h^T take 1.0 as their scale
      RNNT_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)a_y, 0x3f800000, (size_t)D * H, s));
    } else if ((rc = hp_colmax(d->y, M, (int64_t)D * H, (int64_t)D * H, a_y, s))) return rc;
    if (gru) {  // hidden-side gate gradients differ from the input-side ones in the n gate: their own transposed planes
      if (colmax_done) a_dgc = a_y + (int64_t)D * H;   // left there by the v5 recurrence
      else if ((rc = hp_colmax(ghid, M, N4, N4, a_dgc, s))) return rc;
      if ((rc = hp_split_t(ghid, N4, Mv, N4, M, 0, a_dgc, w.hp_dgt, s, ridx))) return rc;
    }
    HpProblem pr[HP_GROUP_MAX];
    int npr = 0;
    if (grouped) pr[npr++] = HpProblem{w.hp_dgt, a_dgc, w.hp_x, a_x, N4, I, Mv, w.wp, I, 0u};
    for (int dir = 0; dir < D; ++dir) {
      // h_prev of frame t is y[t-1] (forward direction) / y[t+1] (reverse): plane row j, index k = y[k -/+ B][dir*H + j], zero outside
      char* yt = w.hp_yt + (size_t)dir * hp_plane_bytes(H, M);
      // (ragged batches: frame (t, b) of the packed contraction takes y of padded row (t -/+ 1, b), which is a valid frame or holds 0)
      if ((rc = hp_split_t(d->y + (int64_t)dir * H, H, Mv, (int64_t)D * H, M, dir == 0 ? -B : B, a_y + (int64_t)dir * H, yt, s, ridx))) return rc;
      const char* ag = w.hp_dgt + (size_t)dir * 4 * H * (size_t)ceil_div(Mv, 32) * 128;
      if (grouped) {
        pr[npr++] = HpProblem{ag, a_dgc + (int64_t)dir * 4 * H, yt, a_y + (int64_t)dir * H, 4 * H, H, Mv, w.dwhh + (int64_t)dir * 4 * H * H, H, 0u};
        continue;
      }
      if ((rc = hp_gemm(ag, a_dgc + (int64_t)dir * 4 * H, yt, a_y + (int64_t)dir * H, 4 * H, H, Mv, w.dwhh + (int64_t)dir * 4 * H * H, 1, H, 0,
                        nullptr, 0, w.scratch, w.scratch_bytes, s)))
        return rc;
    }
    if (grouped) {
      if ((rc = hp_gemm_grouped(pr, npr, xcd_skip, reinterpret_cast<unsigned*>(w.scratch), (char*)w.scratch + HPQ_HEADER_BYTES,
                                w.scratch_bytes - HPQ_HEADER_BYTES, s, k.status)))
        return rc;
      const long per = (long)4 * H * I;
      hipLaunchKernelGGL(unpermute_w_kernel, dim3((unsigned)ceil_div(per, 256), D), dim3(256), 0, s, w.wp, H, I, per, ngate,
                         bd->dw_ih[0], D > 1 ? bd->dw_ih[1] : bd->dw_ih[0], acc);
      RNNT_CHECK_LAUNCH();
    }
  } else
  for (int dir = 0; dir < D; ++dir) {
    rnnt_gemm_desc g = {};
    g.M = 4 * H; g.N = H; g.K = (int64_t)(T - 1) * B;
    const int64_t shift_g = dir == 0 ? (int64_t)B * N4 : 0;           // dG rows t = 1..T-1 | 0..T-2
    const int64_t shift_y = dir == 0 ? 0 : (int64_t)B * D * H;        // y  rows t = 0..T-2 | 1..T-1
    g.A = ghid + shift_g + (int64_t)dir * 4 * H; g.a_mc = 1; g.a_sk = N4; g.a_div = 1;
    g.B = d->y + shift_y + (int64_t)dir * H; g.b_sn = 1; g.b_sk = (int64_t)D * H;
    g.C = w.dwhh + (int64_t)dir * 4 * H * H; g.c_div = 1; g.c_so = H; g.c_si = 0;
    g.workspace = w.scratch; g.workspace_bytes = w.scratch_bytes;
    if (g.K > 0) {
      if ((rc = rnnt_hip_gemm_f32(&g, s))) return rc;
    } else {
      RNNT_CHECK_HIP(hipMemsetAsync(g.C, 0, (size_t)4 * H * H * 4, s));
    }
  }
  {
    const long per = (long)4 * H * H;
    hipLaunchKernelGGL(unpermute_w_kernel, dim3((unsigned)ceil_div(per, 256), D), dim3(256), 0, s, w.dwhh, H, H, per, ngate,
                       bd->dw_hh[0], D > 1 ? bd->dw_hh[1] : bd->dw_hh[0], acc);
    RNNT_CHECK_LAUNCH();
  }
  // 5. bias gradient = column sums of dG, un-permuted (the v4 recurrence already summed its own cells over time)
  if (fused_db) {
    hipLaunchKernelGGL(db_reduce_kernel, dim3((unsigned)ceil_div(4 * H, 256), D), dim3(256), 0, s, k.dbp, db_rows, H, ngate, bd->db[0],
                       D > 1 ? bd->db[1] : bd->db[0], acc, twin0, twin1);
    RNNT_CHECK_LAUNCH();
    if (gru) {
      hipLaunchKernelGGL(db_reduce_kernel, dim3((unsigned)ceil_div(4 * H, 256), D), dim3(256), 0, s, k.dbhp, db_rows, H, ngate, bd->db_hh[0],
                         D > 1 ? bd->db_hh[1] : bd->db_hh[0], acc);
      RNNT_CHECK_LAUNCH();
    }
  } else {
    if ((rc = launch_colsum(d->gates, (long)M, (long)N4, (long)N4, w.bp, w.scratch, w.scratch_bytes, s))) return rc;
    hipLaunchKernelGGL(unpermute_w_kernel, dim3((unsigned)ceil_div(4 * H, 256), D), dim3(256), 0, s, w.bp, H, 1, (long)4 * H, ngate,
                       bd->db[0], D > 1 ? bd->db[1] : bd->db[0], acc, twin0, twin1);
    RNNT_CHECK_LAUNCH();
    if (gru) {  // b_hh sees the hidden-side gradients (n gate scaled by r)
      if ((rc = launch_colsum(ghid, (long)M, (long)N4, (long)N4, w.bp, w.scratch, w.scratch_bytes, s))) return rc;
      hipLaunchKernelGGL(unpermute_w_kernel, dim3((unsigned)ceil_div(4 * H, 256), D), dim3(256), 0, s, w.bp, H, 1, (long)4 * H, ngate,
                         bd->db_hh[0], D > 1 ? bd->db_hh[1] : bd->db_hh[0], acc);
      RNNT_CHECK_LAUNCH();
    }
  }
  return RNNT_OK;
}

extern "C" int rnnt_hip_lstm_check(const void* workspace, void* stream) {
  // word 0 of the workspace is the persistent kernels' status word (0 = ok, 1 = an inter-CU wait gave up)
  RNNT_CHECK_ARG(workspace != nullptr, "lstm_check: null workspace");
  unsigned st = 0;
  RNNT_CHECK_HIP(hipMemcpyAsync(&st, workspace, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
  RNNT_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (st != 0) {
    set_error("persistent LSTM kernel abandoned an inter-workgroup wait (status %u)", st);
    return RNNT_ERR_TIMEOUT;
  }
  return RNNT_OK;
}

extern "C" int rnnt_hip_lstm_debug_read(const void* workspace, int32_t T, int32_t B, int32_t I, int32_t H, int32_t D,
                                        uint64_t* out, int32_t nwg, void* stream) {
  Plan pl;
  int cus = device_cus();
  RNNT_CHECK_ARG(workspace && out && nwg >= 1 && nwg <= 512, "lstm_debug_read: bad arguments");
  RNNT_CHECK_ARG(make_plan(B, H, D, cus, &pl), "lstm_debug_read: unsupported shape");
  const LstmWs w = carve_lstm(const_cast<void*>(workspace), T, B, I, H, D, pl);
  RNNT_CHECK_HIP(hipMemcpyAsync(out, w.dbg, (size_t)nwg * 8 * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  RNNT_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return RNNT_OK;
}

extern "C" size_t rnnt_hip_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M < 0 || N < 1) return 0;
  return (size_t)colsum_chunks((long)M, (long)N) * (size_t)N * 4;
}

extern "C" int rnnt_hip_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  RNNT_CHECK_ARG(X && out && M >= 0 && N >= 1 && ld >= N, "colsum: bad arguments");
  return launch_colsum(X, (long)M, (long)N, (long)ld, out, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int rnnt_hip_colsum_f32_acc(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  RNNT_CHECK_ARG(X && out && M >= 0 && N >= 1 && ld >= N, "colsum: bad arguments");
  return launch_colsum(X, (long)M, (long)N, (long)ld, out, workspace, workspace_bytes, (hipStream_t)stream, 1);
}

extern "C" int rnnt_hip_embedding_fwd(const float* W, const int64_t* idx, int64_t M, int32_t H, int32_t V, float* out,
                                      void* stream) {
  RNNT_CHECK_ARG(W && idx && out && M >= 0 && H >= 1 && V >= 1, "embedding_fwd: bad arguments");
  if (M == 0) return RNNT_OK;
  const long blocks = ceil_div(M * H, 256);
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, W,
                     (const long*)idx, (long)M, H, V, out);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

static int embedding_bwd_impl(const float* dE, const int64_t* idx, int64_t M, int32_t H, int32_t V, int64_t padding_idx,
                              float* dW, int accumulate, void* stream) {
  RNNT_CHECK_ARG(dE && idx && dW && M >= 0 && H >= 1 && V >= 1, "embedding_bwd: bad arguments");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, dE, (const long*)idx, (long)M, H, V,
                     (long)padding_idx, dW, accumulate);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}
extern "C" int rnnt_hip_embedding_bwd(const float* dE, const int64_t* idx, int64_t M, int32_t H, int32_t V,
                                      int64_t padding_idx, float* dW, void* stream) {
  return embedding_bwd_impl(dE, idx, M, H, V, padding_idx, dW, 0, stream);
}
extern "C" int rnnt_hip_embedding_bwd_acc(const float* dE, const int64_t* idx, int64_t M, int32_t H, int32_t V,
                                          int64_t padding_idx, float* dW, void* stream) {
  return embedding_bwd_impl(dE, idx, M, H, V, padding_idx, dW, 1, stream);
}
