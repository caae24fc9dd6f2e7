// v5 persistent recurrences: tagged-payload exchange (no flag round), f16 matrix cores on half-pair operands.
//
// Same decomposition as lstm.hip's v3 / v4 register forms — sync groups = direction x batch slice (<= 16 rows), a workgroup owns
// 16 hidden units (64 gate columns), W_hh stationary in registers, K split over the 4 waves — with two changes that cut the
// per-timestep dependency chain (profiles/r01_lstm_phase_cycles_v3_v4.txt: 7.1 k cycles forward, of which 2.95 k were the
// flag protocol and 1.5 k the six bf16 piece products):
//
//   1. The exchanged value IS the flag.  A hidden value travels as ONE dword = (hi fp16 | lo fp16 << 16) with the generation bit of
//      its timestep in the lowest mantissa bit of lo; the producer's 4-byte store is the publication (no vmcnt drain, no workgroup
//      barrier, no flag store) and a consumer wave polls exactly the 128 k x 16 rows it multiplies (8 producers), re-loading
//      until every dword carries the expected generation.  One L2 round trip after the producer's store lands, instead of
//      drain -> barrier -> flag -> poll -> barrier -> gather.  Buffers alternate by step parity; a slot is overwritten two steps later,
//      which its producer can only reach after every consumer has published the step in between (data dependency = the old
//      flag order), so a reader sees the wanted generation or the one two steps older, never a mix it cannot tell apart.
//   2. h . W_hh^T on v_mfma_f32_16x16x32_f16 with half-pair operands (gemm_hp.hip's arithmetic): h in (-1, 1) scaled by 2^14,
//      W_hh scaled by a power of two from the workgroup's own slice maximum; 3 products (lo.hi + hi.lo + hi.hi) instead of 6.
//
//   3. Exchange images are laid out [k / 4][row][k % 4] (forward: packed h; backward: per producer, [unit / 4][row][unit % 4]): the
//      lanes of one gather instruction read, and the lanes of one publishing instruction write, ONE contiguous block instead of a
//      16-byte piece per row 4 * Kp bytes apart (the poll traffic is re-issued every round, so its request count is what the
//      hand-off costs: c2 forward 8.6 -> 8.0 ms, backward 10.0 -> 9.5 ms per training step).
//   4. A step's stash stores and the next step's stash loads are issued AFTER the arrival of this step's operands, not behind the
//      publication: between a publication and the arrival of the next operands a CU's memory queue holds the polls only.
//
// Correctness never depends on placement: stores are sc1 write-through unless the group is VERIFIED to sit on one XCD (then
// plain stores stay in that L2), loads are always sc1 (bypass L1).  All spins are bounded and raise the status word.
#include "lstm_shared.hpp"
#include <string.h>

namespace rnnt {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned TAG_MASK = 0x00010000u;   // lowest mantissa bit of the lo half

__device__ __forceinline__ unsigned pack_hp(float v, unsigned tag) {
  // v already scaled (|v| < 2^15): hi = fp16_rn(v), lo = fp16_rn(v - hi) with its last bit replaced by the generation tag
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  const unsigned h = (unsigned)__builtin_bit_cast(unsigned short, hi), l = (unsigned)__builtin_bit_cast(unsigned short, lo);
  return h | ((l & 0xfffeu) << 16) | (tag << 16);
}

// 8 fp32 weights (consecutive k) -> hi / lo f16x8 fragments, scaled
__device__ __forceinline__ void split8h(const f32x4& a, const f32x4& b, float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (e < 4 ? a[e] : b[e - 4]) * scale;
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
}

// Steps a sync group runs: T, or — ragged batches, p.gbound — the longest of its rows (uniform over the group's workgroups: every
// member reads the same <= 16 lengths).  Frames beyond it are never touched by the group: the host zero-fills y and gathers the valid
// rows of everything else (rnnt_lstm_desc.row_idx).
__device__ __forceinline__ int group_steps(const LstmK& p, int b0) {
  if (!p.gbound) return p.T;
  int m = 1;
  for (int r = 0; r < p.Bg && b0 + r < p.B; ++r) m = max(m, min(p.lens[b0 + r], p.T));
  return __builtin_amdgcn_readfirstlane(m);
}

// Waits until every lane's predicate holds (wave-level), re-running `load_and_check` (which (re)issues the lane's loads and returns
// whether all its dwords carry the expected tag).  Bounded: gives up after SPIN_LIMIT_TICKS or when another workgroup raised the
// status word, raising it itself.  Returns false on abort.
template <typename F>
__device__ __forceinline__ bool poll_tagged(F&& load_and_check, unsigned* status) {
  if (__all(load_and_check())) return true;
  const unsigned long long t0 = wall_clock64();
  unsigned spins = 0;
  while (true) {
    __builtin_amdgcn_s_sleep(1);   // (no sleep, or 2 / 4 / 8 times longer: within +-0.5 % of this per c2 step — the loop's pace is not the limit)
    if (__all(load_and_check())) return true;
    if ((++spins & 63u) == 0u) {
      const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (st != 0u || wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
        if ((threadIdx.x & 63) == 0 && st == 0u) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// ================================================================================================
// forward.  dynamic LDS: part[2 parities][NWV waves][MB blocks][64] f32x4 | wmax[8] | abort | pubs[16][HS]
// NKS: 32-deep k-steps per wave (Kp = NWV * 32 * NKS >= H, zero padded).  CELL: 0 LSTM, 1 GRU, 2 tanh Elman RNN.
// NWV x MB: 4 x 4 (H = 128..512: 16 units per workgroup), 8 x 5 (H = 640: 20 units per workgroup so that a sync group has 32
// members and fits one XCD; K padded to 768) or 8 x 4 (H = 768 / 1024: NKS = 3 / 4, 48 / 64 workgroups per group).  Wave w < MB owns gate-column block w (one cell per lane).
// ================================================================================================
template <int NKS, int CELL, int NWV = 4, int MB = 4>
__global__ void __launch_bounds__(64 * NWV) lstm_fwd5_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int HS = 4 * MB, Kw = 32 * NKS;
  static_assert(MB <= NWV, "one owner wave per gate-column block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  float* wmax = reinterpret_cast<float*>(part + 2 * NWV * MB * 64);
  int* abort_lds = reinterpret_cast<int*>(wmax + 8);
  unsigned* pubs = reinterpret_cast<unsigned*>(abort_lds + 4);   // [16 rows][HS units] packed dwords (write-through path only)

  const int H = p.H, B = p.B, D = p.D, Kp = p.Kp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the per-wave role tests below become scalar branches
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;
  const int lrow = lane & 15, lq = lane >> 4;
  const int TT = p.T;                       // padded frames: strides of the stash
  const int T = group_steps(p, b0);         // steps this sync group runs (ragged batches: the longest of its rows)

  // W_hh slice: lane -> gate column 16*mb + lrow (gate lrow&3 of unit 4*mb + (lrow>>2)), k = wave*Kw + 32*ks + 8*lq + e
  f16x8 whi[MB][NKS], wlo[MB][NKS];
  float out_scale;
  {
    const float* W = p.w_hh[d];
    const int gate = lrow & 3;
    f32x4 raw[MB][NKS][2];
    float m = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float* row = W + (long)(gate * H + j0 + 4 * mb + (lrow >> 2)) * H;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int k = wave * Kw + 32 * ks + 8 * lq;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (gate < NGATE && k < H) {   // H is a multiple of 8: an 8-chunk is inside or outside as a whole
          lo = *reinterpret_cast<const f32x4*>(row + k);
          hi = *reinterpret_cast<const f32x4*>(row + k + 4);
        }
        raw[mb][ks][0] = lo;
        raw[mb][ks][1] = hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) wmax[wave] = m;
    if (tid == 0) *abort_lds = 0;
    __syncthreads();
    m = wmax[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmaxf(m, wmax[w]);
    // power-of-two scale bringing the slice maximum into [2^14, 2^15)
    int eb = (int)((__float_as_uint(m) >> 23) & 255u);
    eb = eb < 15 ? 15 : eb;
    const float wscale = m > 0.f ? __uint_as_float((unsigned)(268 - eb) << 23) : 1.f;
    out_scale = (m > 0.f ? __uint_as_float((unsigned)(eb - 14) << 23) : 1.f) * 6.103515625e-05f;  // / wscale / 2^14
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) split8h(raw[mb][ks][0], raw[mb][ks][1], wscale, whi[mb][ks], wlo[mb][ks]);
  }

  const int NBR = 4 * ((p.Bg + 3) / 4);  // exchange rows of the group (host allocation)
  const long hx_bytes = (long)NBR * Kp * 4;  // one (parity, group) image: [row][Kp] packed dwords
  __amdgpu_buffer_rsrc_t hx_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    hx_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.hx) + ((long)par * NG + gid) * hx_bytes, 0, (int)hx_bytes,
                                                     RSRC_FLAGS);

  // one cell per lane of waves 0..MB-1: unit 4*wave + lq of this workgroup, batch row lrow of this group (further waves only
  // contribute their K-slice of the product)
  const bool ownw = wave < MB;
  const int brow = lrow;
  const int ob = b0 + brow, oj = j0 + 4 * (ownw ? wave : 0) + lq;
  const bool inrow = brow < NBR;                       // exchange row of the group (gather side: every wave)
  const bool valid = ownw && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float c_state = 0.f;
  const float bhn = (CELL == 1 && ownw) ? p.b_hh[d][2 * H + oj] : 0.f;
  // stash / prefetch through buffer resources with per-lane byte offsets: inactive lanes carry an out-of-range offset (loads
  // return 0, stores are dropped), so the loop body has NO divergent branch around a memory operation and hipcc can count
  // exactly how many younger operations may stay in flight when it waits for the gathered operands (a vmcnt(0) there would also
  // wait for this step's stash stores)
  constexpr int OOB = 0x7ffffff0;
  const int t_first = (d == 0) ? 0 : T - 1;
  const int tdir = (d == 0) ? 1 : -1;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)((long)TT * B * D * 4 * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.cst, 0, CELL == 0 ? (int)((long)D * TT * B * H * 4) : 0, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long)TT * B * D * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t yd_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.ydrop, 0, p.ydrop ? (int)((long)TT * B * D * H * 4) : 0, RSRC_FLAGS);
  int g_off = valid ? (int)(((((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj) * 4) : OOB;
  int c_off = valid ? (int)((((((long)d * TT + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3)) * 4) : OOB;
  int y_off = valid ? (int)(((((long)t_first * B + ob) * D + d) * H + oj) * 4) : OOB;
  const int g_step = valid ? tdir * B * D * 4 * H * 4 : 0, c_step = valid ? tdir * H * B * 4 : 0, y_step = valid ? tdir * B * D * H * 4 : 0;
  // exchange image laid out [k / 4][row][k % 4]: the 16-byte chunks of one k-quad of all rows are adjacent, so the lanes of one
  // gather instruction (rows 0..NBR-1 of one k-quad) read ONE contiguous NBR x 16 bytes and an owner wave's publication (4 units x
  // NBR rows) is one contiguous block — instead of NBR requests 4 * Kp bytes apart
  const int hx_off = (ownw && inrow) ? (((oj >> 2) * NBR + brow) * 4 + (oj & 3)) * 4 : OOB;
  const int gat_off = inrow ? ((((wave * Kw + 8 * lq) >> 2) * NBR) + brow) * 16 : 0x7ffffff0;  // rows beyond the group read 0 and are not checked
  const int gat_ks = 8 * NBR * 16, gat_hi = NBR * 16;   // next 32-deep k-step / second k-quad of the lane's 8 values
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  u32x4 raw[NKS][2];   // the next step's operand dwords: loads are issued right behind this step's publication
  // k-steps beyond H (zero padding of K, H = 640 form) have no producer: not loaded (zeros), not checked
  auto issue_gather = [&](int s_next) {
    const __amdgpu_buffer_rsrc_t src = hx_rsrc[(s_next - 1) & 1];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int off = (wave * Kw + 32 * ks < H) ? gat_off + gat_ks * ks : OOB;
      raw[ks][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, off, 0, AUX_SC1));
      raw[ks][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, off + gat_hi, 0, AUX_SC1));
    }
  };
  auto tags_ok = [&](unsigned want) -> bool {
    unsigned bad = 0;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      if (wave * Kw + 32 * ks < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= (raw[ks][0][e] ^ want) | (raw[ks][1][e] ^ want);
      }
    return !inrow || (bad & TAG_MASK) == 0u;
  };
  // gate pre-activations (W_ih x + b, written by the input-projection GEMM).  H = 640 form: fetched TWO steps ahead into one of three
  // register sets (loop body instantiated three times, so no register is moved while its load is in flight): c5 17.9 -> 17.4 ms of
  // forward recurrences per step.  Other forms: one step ahead (two steps ahead measured 6.39 vs 6.22 ms at c2, equal at c3).
  constexpr int AHEAD = MB == 5 ? 2 : 1;
  f32x4 xpA = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_off, 0, 0));
  f32x4 xpB = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, AHEAD == 2 && T > 1 ? g_off + g_step : OOB, 0, 0));
  f32x4 xpC = {0.f, 0.f, 0.f, 0.f};
  // the previous step's stash, kept in registers until this step's operands have arrived: between a publication and the arrival
  // of the next operands the CU's memory queue holds the operand loads only
  i32x4 st_g = {0, 0, 0, 0};
  int st_c = 0, st_y = 0, st_yd = 0, st_goff = OOB, st_coff = OOB, st_yoff = OOB;
  auto flush_stash = [&]() {
    __builtin_amdgcn_raw_buffer_store_b128(st_g, g_rsrc, st_goff, 0, 0);
    if constexpr (CELL == 0) __builtin_amdgcn_raw_buffer_store_b32(st_c, c_rsrc, st_coff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(st_y, y_rsrc, st_yoff, 0, 0);
    if (p.ydrop) __builtin_amdgcn_raw_buffer_store_b32(st_yd, yd_rsrc, st_yoff, 0, 0);
  };
  auto step = [&](const int s, const f32x4& xp, f32x4& xp_ld) -> bool {
    const int t = (d == 0) ? s : T - 1 - s;
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    DBG_STAMP(0);
    bool ok = true;
    if (s > 0) {
      const unsigned want = ((((unsigned)(s - 1) >> 1) & 1u) ^ 1u) << 16;
      if (!__all(tags_ok(want))) {   // first round was issued behind the previous step's publication
        ok = poll_tagged([&]() -> bool { issue_gather(s); return tags_ok(want); }, p.status);
      }
      DBG_STAMP(1);
    }
    if (s > 0) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // de-interleave 8 packed dwords into the hi / lo operand fragments (k order kept), tag bit cleared
        u32x4 h4, l4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned d0 = j < 2 ? raw[ks][0][2 * j] : raw[ks][1][2 * j - 4], d1 = j < 2 ? raw[ks][0][2 * j + 1] : raw[ks][1][2 * j - 3];
          h4[j] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);                    // {d1.lo16, d0.lo16}
          l4[j] = __builtin_amdgcn_perm(d1, d0, 0x07060302u) & 0xfffefffeu;      // {d1.hi16, d0.hi16}
        }
        const f16x8 hh = __builtin_bit_cast(f16x8, h4), hl = __builtin_bit_cast(f16x8, l4);
        // (block-major: three dependent MFMAs per accumulator in a row.  Term-major order — four independent accumulators between two
        //  MFMAs of a chain — measured 1 % SLOWER in paired runs, 5.97 vs 5.92 ms of forward recurrences per c2 step: the SIMD's other
        //  wave already fills the dependent-issue gaps)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[mb][ks], hh, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb][ks], hl, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb][ks], hh, acc[mb], 0, 0, 0);
        }
      }
      DBG_STAMP(2);
    }
    // the previous step's stash and the next step's gate pre-activations: issued BEHIND this step's MFMAs (they execute while these
    // instructions issue; in front of them — or between the two k-steps' MFMAs — the same five VMEM issues cost 760 cycles per step)
    xp_ld = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, s + AHEAD < T ? g_off + AHEAD * g_step : OOB, 0, 0));
    flush_stash();
    if (!ok) *abort_lds = 1;   // benign race: any wave that gave up makes the whole workgroup leave after the barrier
    f32x4* pp = part + (s & 1) * (NWV * MB * 64);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) pp[(wave * MB + mb) * 64 + lane] = acc[mb];
    __syncthreads();
    DBG_STAMP(3);   // partial write + barrier (includes the skew between this workgroup's waves)
    if (*abort_lds != 0) return false;
    const int ownb = ownw ? wave : 0;
    f32x4 rec = pp[ownb * 64 + lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w) rec += pp[(w * MB + ownb) * 64 + lane];
    rec *= out_scale;
    const bool active = valid && t < olen;
    float hval = 0.f;
    f32x4 gact = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      if constexpr (CELL == 0) {
        const f32x4 g4 = xp + rec;
        const float ig = sigmoid_hw(g4[0]), fg = sigmoid_hw(g4[1]), gg = tanh_hw(g4[2]), og = sigmoid_hw(g4[3]);
        c_state = fg * c_state + ig * gg;
        hval = og * tanh_hw(c_state);
        gact = (f32x4){ig, fg, gg, og};
      } else if constexpr (CELL == 1) {
        const float rg = sigmoid_hw(xp[0] + rec[0]), zg = sigmoid_hw(xp[1] + rec[1]);
        const float hn = rec[2] + bhn;
        const float ng = tanh_hw(xp[2] + rg * hn);
        hval = (1.f - zg) * ng + zg * c_state;
        c_state = hval;
        gact = (f32x4){rg, zg, ng, hn};
      } else {
        hval = tanh_hw(xp[0] + rec[0]);
        c_state = hval;
        gact = (f32x4){hval, 0.f, 0.f, 0.f};
      }
    } else {
      c_state = 0.f;
    }
    {  // the publication: ONE dword = (hi | lo << 16) of h * 2^14, generation bit in lo's last mantissa bit
      const unsigned tag = (((unsigned)s >> 1) & 1u) ^ 1u;
      const int v = (int)pack_hp(hval * 16384.f, tag);
      if constexpr (LOCAL) {
        __builtin_amdgcn_raw_buffer_store_b32(v, hx_rsrc[s & 1], hx_off, 0, 0);   // stays in the group's L2
      } else {
        // group spans XCDs: write-through stores, and a 4-byte sc1 store is one fabric write each (12x the time per byte of a
        // 16-byte one): gather the workgroup's 16 rows x HS units in LDS and let wave 0 store them as granules of 16 bytes
        if (ownw) pubs[brow * HS + 4 * wave + lq] = (unsigned)v;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
          for (int gi = lane; gi < 16 * MB; gi += 64) {
            const int r = gi % 16, q = gi / 16;   // consecutive lanes -> consecutive rows of one k-quad: contiguous
            const i32x4 gran = *reinterpret_cast<const i32x4*>(pubs + r * HS + 4 * q);
            __builtin_amdgcn_raw_buffer_store_b128(gran, hx_rsrc[s & 1], r < NBR ? (((j0 >> 2) + q) * NBR + r) * 16 : OOB, 0, AUX_SC1);
          }
        }
      }
    }
    DBG_STAMP(4);   // reduce + cell math + publication
    // first poll round of the next step (unused after the last step); this step's stash stays in registers until it has arrived.
    // (A pause in front of it — s_sleep 6 / 8 — gains 2 % at c2 and c5 and loses 1-3 % at c3: not taken.)
    for (int i = 0; i < (p.pause & 255); ++i) __builtin_amdgcn_s_sleep(1);
    issue_gather(s + 1);
    st_g = __builtin_bit_cast(i32x4, gact);
    st_c = __builtin_bit_cast(int, c_state);
    st_y = __builtin_bit_cast(int, hval);
    if (p.ydrop) st_yd = __builtin_bit_cast(int, (hash_u32(p.seed, (unsigned long long)(unsigned)(y_off >> 2)) >= p.drop_thresh) ? hval * p.keep_scale : 0.f);
    st_goff = g_off; st_coff = c_off; st_yoff = y_off;
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);
    return true;
  };
  if constexpr (AHEAD == 2) {
    for (int s = 0; s < T; s += 3) {
      if (!step(s, xpA, xpC)) return false;
      if (s + 1 < T && !step(s + 1, xpB, xpA)) return false;
      if (s + 2 < T && !step(s + 2, xpC, xpB)) return false;
    }
  } else {
    for (int s = 0; s < T; ++s) {
      if (!step(s, xpA, xpB)) return false;
      xpA = xpB;   // (one register after coalescing: the load is issued behind the last use of xpA)
    }
  }
  flush_stash();
  return true;
  };
  const bool okrun = local ? run(std::true_type{}) : run(std::false_type{});
  if (!okrun) return;
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}


// ================================================================================================
// backward ("scatter form" of lstm.hip's v4 with the v5 exchange and arithmetic).
//   A workgroup multiplies ITS OWN 64 gate columns of dG_t (16 rows x 64, produced by its own cell math) by its 64 rows of
//   W_hh and publishes the partial dh_{t-1} for ALL hidden units; a consumer sums, for its 16 units, the NC partial slices.
//   Exchange: fp32 partial sums as 16-byte granules (4 units of one row), stored by ONE lane straight from the MFMA output
//   registers, generation bit in the last mantissa bit of elements 0 and 2 (one per 8-byte half); a gathering thread re-loads
//   a granule until both halves carry the wanted generation.  No flag, no drain, no publication barrier.
//   Arithmetic: dG rows scaled per (row, workgroup) by a power of two from the row's maximum over the 64 own columns (LDS
//   ds_max), W_hh by the slice maximum; 3 f16 products; the product is descaled before it is published.
// dynamic LDS: red[NT] f32x4 | dgs[16][DGS_LD] float | rowexp[2][16] | wmax[8] | abort
// NMB: 16-unit output blocks per wave (Kp / 16 / NWV).  NWV x MB: 4 x 4 (H = 128..512), 8 x 5 (H = 640, K padded to 768) or 8 x 4
// (H = 768 / 1024: 48 / 64 producers per group, NCMAX = 64; such a group spans XCDs and runs the write-through exchange).
// ================================================================================================
template <int NMB, int BQ, int CELL, int NWV = 4, int MB = 4, int NCMAX = 32>
__global__ void __launch_bounds__(64 * NWV) lstm_bwd5_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int HS = 4 * MB, UQ = MB, NT = 64 * NWV;
  constexpr int KSB = (16 * MB + 31) / 32;   // 32-deep k-steps over the own gate columns (zero padded)
  constexpr int NBR = 4 * BQ;           // exchange rows of the group
  constexpr int DGS_LD = 32 * KSB + 4;
  constexpr int OOB = 0x7ffffff0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* dgs = reinterpret_cast<float*>(red + NT);
  unsigned* rowexp = reinterpret_cast<unsigned*>(dgs + 16 * DGS_LD);
  float* wmax = reinterpret_cast<float*>(rowexp + 32);
  int* abort_lds = reinterpret_cast<int*>(wmax + 8);

  const int H = p.H, B = p.B, D = p.D, Kp = p.Kp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;
  const int lrow = lane & 15, lq = lane >> 4;
  const int TT = p.T;                       // padded frames: strides of the stash
  const int T = group_steps(p, b0);         // steps this sync group runs (ragged batches: the longest of its rows)

  // W_hh slice as the A operand: lane -> output unit u = 16*(wave*NMB + mb) + lrow (a COLUMN of W_hh),
  // k = 32*ks + 8*lq + e = own gate column 4*unit + gate  ->  W_hh[gate*H + j0 + (k>>2)][u]
  f16x8 whi[NMB][KSB], wlo[NMB][KSB];
  float w_inv;
  {
    const float* W = p.w_hh[d];
    f32x4 raw[NMB][KSB][2];
    float m = 0.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int u = 16 * (wave * NMB + mb) + lrow;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const int c = 32 * ks + 8 * lq;   // own gate column, a multiple of 8: units c>>2 and (c>>2)+1, gates 0..3 each
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (u < H) {   // output units beyond H: zero padding of K (H = 640 form)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (e < NGATE) {
              if ((c >> 2) < HS) lo[e] = W[(long)(e * H + j0 + (c >> 2)) * H + u];
              if ((c >> 2) + 1 < HS) hi[e] = W[(long)(e * H + j0 + (c >> 2) + 1) * H + u];
            }
          }
        }
        raw[mb][ks][0] = lo;
        raw[mb][ks][1] = hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) wmax[wave] = m;
    if (tid == 0) *abort_lds = 0;
    if (tid < 32) rowexp[tid] = 0u;
    for (int i = tid; i < 16 * DGS_LD; i += NT) dgs[i] = 0.f;
    __syncthreads();
    m = wmax[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmaxf(m, wmax[w]);
    int eb = (int)((__float_as_uint(m) >> 23) & 255u);
    eb = eb < 15 ? 15 : eb;
    const float wscale = m > 0.f ? __uint_as_float((unsigned)(268 - eb) << 23) : 1.f;
    w_inv = m > 0.f ? __uint_as_float((unsigned)(eb - 14) << 23) : 1.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) split8h(raw[mb][ks][0], raw[mb][ks][1], wscale, whi[mb][ks], wlo[mb][ks]);
  }

  const long px_bytes = (long)p.NC * NBR * Kp * 4;  // one (parity, group) image: [producer][row][Kp] fp32
  __amdgpu_buffer_rsrc_t px_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    px_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.hx) + ((long)par * NG + gid) * px_bytes, 0, (int)px_bytes, RSRC_FLAGS);

  // cell owners: tid = ((obq*UQ + uq)*4 + i)*4 + j -> unit j0 + 4*uq + i, batch row 4*obq + j
  const bool owner = tid < BQ * HS * 4;
  const int ojb = tid & 3, oi = (tid >> 2) & 3, ouq = (tid >> 4) % UQ, obq = (tid >> 4) / UQ;
  const int brow = 4 * obq + ojb, ob = b0 + brow, oj = j0 + 4 * ouq + oi;
  const bool valid = owner && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float dc_carry = 0.f;
  f32x4 db_acc = {0.f, 0.f, 0.f, 0.f}, dbh_acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 cmx = {0.f, 0.f, 0.f, 0.f}, cmxh = {0.f, 0.f, 0.f, 0.f};   // running |dG| maxima of this cell's 4 gate columns
  const int t_first = (d == 0) ? T - 1 : 0;
  const int tdir = (d == 0) ? -1 : 1;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)((long)TT * B * D * 4 * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, CELL == 1 ? (int)((long)TT * B * D * 4 * H * 4) : 0, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.cst, 0, CELL == 0 ? (int)((long)D * TT * B * H * 4) : 0, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long)TT * B * D * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((long)TT * B * D * H * 4), RSRC_FLAGS);
  int g_off = valid ? (int)(((((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj) * 4) : OOB;
  int c_off = valid ? (int)((((((long)d * TT + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3)) * 4) : OOB;
  int y_off = valid ? (int)(((((long)t_first * B + ob) * D + d) * H + oj) * 4) : OOB;
  const int g_step = valid ? tdir * B * D * 4 * H * 4 : 0, c_step = valid ? tdir * H * B * 4 : 0, y_step = valid ? tdir * B * D * H * 4 : 0;

  // gather: thread -> (row, unit quad) pair gpr and producer class gq; it sums the partial slices of producers gq, gq + NQ, ...
  // gathering threads: all of them, except in the H = 512 form with 8 waves, where the first 4 waves gather (as in the 4-wave form) and
  // the other 4 only share the MFMA + publication phase
  constexpr int NTG = (NWV == 8 && MB == 4 && NCMAX == 32) ? 256 : NT;
  constexpr int NPAIR = NBR * UQ, NQ = NTG / NPAIR, NLD = (NCMAX + NQ - 1) / NQ;   // NC <= NCMAX; threads beyond NQ * NPAIR only help elsewhere
  const int gq = tid / NPAIR, gpr = tid % NPAIR;
  const bool gact = gq < NQ;
  // one producer's image laid out [unit / 4][row][unit % 4] (16-byte granules of one unit quad of all rows adjacent): a publishing
  // instruction (4 quads x NBR rows) writes one contiguous block, the gathering threads of one producer read one contiguous block
  const int grow = gpr % NBR, guq = gpr / NBR;
  const int gat_base = (((gq * (Kp >> 2)) + (j0 >> 2) + guq) * NBR + grow) * 16;
  const int pub_base = lrow < NBR ? ((wg * (Kp >> 2) + 4 * wave * NMB + lq) * NBR + lrow) * 16 : OOB;
  constexpr int pub_mb = 4 * NBR * 16;
  auto pair_index = [](int row, int uq) { return uq * NBR + row; };
  const int gat_step = NQ * NBR * Kp * 4;
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  u32x4 gr[NLD];
  auto issue_gather = [&](int s_next) {   // partials published at step s_next - 1
    const __amdgpu_buffer_rsrc_t src = px_rsrc[(s_next - 1) & 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool okp = gact && gq + NQ * i < p.NC;
      gr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, okp ? gat_base + i * gat_step : OOB, 0, AUX_SC1));
    }
  };
  auto tags_ok = [&](unsigned want) -> bool {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (gact && gq + NQ * i < p.NC) bad |= (gr[i][0] ^ want) | (gr[i][2] ^ want);
    return (bad & 1u) == 0u;
  };
  // prefetch of step 0
  f32x4 gt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_off, 0, 0));
  float c_t = 0.f, c_p = 0.f;
  if constexpr (CELL == 0) {
    c_t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, c_off, 0, 0));
    c_p = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, T > 1 ? c_off + c_step : OOB, 0, 0));
  } else if constexpr (CELL == 1) {
    c_p = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rsrc, T > 1 ? y_off + y_step : OOB, 0, 0));
  }
  float dyv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, y_off, 0, 0));
  // the previous step's stash (dG, row maxima), kept in registers until this step's partial sums have arrived; the next step's stash
  // reads are issued there too: between a publication and the arrival of the next partial sums the CU's memory queue holds the
  // gather loads only
  i32x4 st_dg = {0, 0, 0, 0}, st_dgh = {0, 0, 0, 0};
  int st_goff = OOB, st_t = 0;
  unsigned st_rm = 0u;
  f32x4 gt_n = {0.f, 0.f, 0.f, 0.f};
  float ct_n = 0.f, cp_n = 0.f, dy_n = 0.f;
  auto flush_stash = [&]() {
    __builtin_amdgcn_raw_buffer_store_b128(st_dg, g_rsrc, st_goff, 0, 0);
    if constexpr (CELL == 1) __builtin_amdgcn_raw_buffer_store_b128(st_dgh, a_rsrc, st_goff, 0, 0);
    if constexpr (CELL != 1) {
      if (p.rowmax && st_rm != 0u) atomicMax(p.rowmax + (long)st_t * B + b0 + tid, st_rm);
    }
  };
  for (int s = 0; s < T; ++s) {
    const int t = (d == 0) ? T - 1 - s : s;
    const bool active = valid && t < olen;
    if (p.ydrop) dyv = (hash_u32(p.seed, (unsigned long long)(unsigned)(y_off >> 2)) >= p.drop_thresh) ? dyv * p.keep_scale : 0.f;
    DBG_STAMP(0);
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (s > 0) {
      const unsigned want = (((unsigned)(s - 1) >> 1) & 1u) ^ 1u;
      if (!__all(tags_ok(want))) ok = poll_tagged([&]() -> bool { issue_gather(s); return tags_ok(want); }, p.status);
      DBG_STAMP(1);
    }
    auto stash_traffic = [&]() {   // the next step's stash reads, the previous step's stash stores (issued inside the MFMA phase below)
      gt_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, s + 1 < T ? g_off + g_step : OOB, 0, 0));
      if constexpr (CELL == 0) {
        ct_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, s + 1 < T ? c_off + c_step : OOB, 0, 0));
        cp_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, s + 2 < T ? c_off + 2 * c_step : OOB, 0, 0));
      } else if constexpr (CELL == 1) {
        cp_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rsrc, s + 2 < T ? y_off + 2 * y_step : OOB, 0, 0));
      }
      dy_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, s + 1 < T ? y_off + y_step : OOB, 0, 0));
      flush_stash();
    };
    if (s > 0) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        u32x4 v = gr[i];
        v[0] &= ~1u;
        v[2] &= ~1u;
        gsum += __builtin_bit_cast(f32x4, v);   // producers beyond NC loaded zeros
      }
    }
    if (!ok) *abort_lds = 1;
    red[tid] = gsum;
    __syncthreads();
    if (*abort_lds != 0) return false;
    DBG_STAMP(2);  // partial sums + barrier
    f32x4 dg4 = {0.f, 0.f, 0.f, 0.f}, dgh4 = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      float dh = active ? dyv : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) dh += red[q * NPAIR + pair_index(brow, ouq)][oi];   // (threads with gq >= NQ wrote zeros: never read)
      if (active) {
        if constexpr (CELL == 0) {
          const float ig = gt[0], fg = gt[1], gg = gt[2], og = gt[3];
          const float tc = tanh_hw(c_t);
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
          dg4 = (f32x4){dh * (1.f - hv * hv), 0.f, 0.f, 0.f};
          dgh4 = dg4;
        }
      } else {
        dc_carry = 0.f;
      }
      db_acc += dg4;
      if constexpr (CELL == 1) dbh_acc += dgh4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cmx[e] = fmaxf(cmx[e], fabsf(dg4[e]));
        if constexpr (CELL == 1) cmxh[e] = fmaxf(cmxh[e], fabsf(dgh4[e]));
      }
      *reinterpret_cast<f32x4*>(&dgs[brow * DGS_LD + 4 * (4 * ouq + oi)]) = dgh4;  // own gate column 4*unit + gate
      const float mx = fmaxf(fmaxf(fabsf(dgh4[0]), fabsf(dgh4[1])), fmaxf(fabsf(dgh4[2]), fabsf(dgh4[3])));
      atomicMax(&rowexp[(s & 1) * 16 + brow], __float_as_uint(mx));   // LDS ds_max_u32: the row's maximum over the 64 own columns
    }
    if (tid < 16) rowexp[((s & 1) ^ 1) * 16 + tid] = 0u;   // the other parity, for the next step
    __syncthreads();
    DBG_STAMP(3);  // cell math
    {
      const unsigned rmax = rowexp[(s & 1) * 16 + lrow];
      int eb = (int)((rmax >> 23) & 255u);
      eb = eb < 15 ? 15 : eb;
      const float gscale = rmax != 0u ? __uint_as_float((unsigned)(268 - eb) << 23) : 1.f;
      const float ginv = (rmax != 0u ? __uint_as_float((unsigned)(eb - 14) << 23) : 1.f) * w_inv;
      f16x8 ghi[KSB], glo[KSB];
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const float* src = dgs + lrow * DGS_LD + 32 * ks + 8 * lq;
        split8h(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), gscale, ghi[ks], glo[ks]);
      }
      const unsigned tag = (((unsigned)s >> 1) & 1u) ^ 1u;
      // groups of 4 output blocks: 4 independent accumulators between dependent MFMAs
      constexpr int GB = (NMB % 4 == 0 && NMB > 4) ? 4 : 2;   // NMB is 2, 4, 6 or 8; at least two groups when NMB >= 4 (stash traffic behind the first)
#pragma unroll
      for (int mb0 = 0; mb0 < NMB; mb0 += GB) {
        f32x4 acc[GB];
#pragma unroll
        for (int j = 0; j < GB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[mb0 + j][ks], ghi[ks], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb0 + j][ks], glo[ks], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb0 + j][ks], ghi[ks], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
          u32x4 out = __builtin_bit_cast(u32x4, acc[j] * ginv);
          out[0] = (out[0] & ~1u) | tag;
          out[2] = (out[2] & ~1u) | tag;
          if constexpr (LOCAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, out), px_rsrc[s & 1], pub_base + pub_mb * (mb0 + j), 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, out), px_rsrc[s & 1], pub_base + pub_mb * (mb0 + j), 0, AUX_SC1);
        }
        // behind the first group's MFMAs and publication stores: these VMEM instructions issue while the second group's MFMAs
        // execute.  (Right behind the poll they delayed the whole local chain: 9.5 -> 8.4 ms of backward recurrences per c2 step.)
        // (behind the LAST group's MFMAs, or behind the whole phase: 9.35 instead of 8.4 ms)
        if (mb0 == 0) stash_traffic();
      }
    }
    DBG_STAMP(4);  // scale + MFMA + publication
    for (int i = 0; i < (p.pause & 255); ++i) __builtin_amdgcn_s_sleep(1);
    issue_gather(s + 1);
    st_dg = __builtin_bit_cast(i32x4, dg4);
    if constexpr (CELL == 1) st_dgh = __builtin_bit_cast(i32x4, dgh4);
    st_goff = g_off;
    st_t = t;
    if constexpr (CELL != 1) {
      st_rm = 0u;
      if (p.rowmax && tid < NBR && tid < p.Bg && b0 + tid < B) st_rm = rowexp[(s & 1) * 16 + tid];   // (cleared a step later)
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    gt = gt_n; c_t = ct_n; c_p = cp_n; dyv = dy_n;
    DBG_STAMP(5);  // gather issue + prefetch + stash
  }
  flush_stash();
  return true;
  };
  const bool okrun = local ? run(std::true_type{}) : run(std::false_type{});
  if (!okrun) return;
  if (owner) {  // one row of the (group, exchange row) table per cell row; summed over rows and groups by db_reduce_kernel
    const long row = (long)gid * NBR + brow;
    *reinterpret_cast<f32x4*>(p.dbp + row * 4 * H + 4 * oj) = db_acc;
    if constexpr (CELL == 1) *reinterpret_cast<f32x4*>(p.dbhp + row * 4 * H + 4 * oj) = dbh_acc;
    if (p.colmax) {   // column maxima of dG for the half-pair planes of the weight-gradient products (max is order-independent)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned a = __float_as_uint(cmx[e]);
        if (a != 0u) atomicMax(p.colmax + (long)d * 4 * H + 4 * oj + e, a);
        if constexpr (CELL == 1) {
          const unsigned b = __float_as_uint(cmxh[e]);
          if (b != 0u) atomicMax(p.colmax_h + (long)d * 4 * H + 4 * oj + e, b);
        }
      }
    }
  }
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

// ================================================================================================
// backward, ONE barrier per step (H = 128..512: 4 waves x 16 units, <= 32 producers per group; lstm_bwd5_kernel keeps the other shapes
// and is the A/B partner: RNNT_LSTM_BWD5_2B=1).  Same decomposition, exchange protocol and arithmetic as lstm_bwd5_kernel; what
// changed is who does what inside the workgroup (profiles/r02_lstm_phase_cycles_v5_final.txt: of 4 521 cycles per step 748 were
// "partial sums + barrier" and 2 021 "scale + MFMA + publication", where every wave converted the whole 16 x 64 dG image to f16 pairs):
//   1. Wave w gathers AND owns exchange rows w*BQ .. w*BQ+BQ-1: lane = class * PPW + (unit quad * BQ + row), class = the producers
//      c, c + NQ, ... the lane sums in registers.  The sum over classes is a butterfly over lane bits (DPP rotations inside a row of
//      16 lanes, v_permlane16/32_swap across rows) — no LDS round trip, no barrier; lanes of class 0..3 then own the cell of unit
//      4 * quad + class.  The exchange image orders a block's granules [row / BQ][quad][row % BQ] so that such a wave still reads one
//      contiguous run per producer and a publishing instruction still writes one contiguous block.
//   2. The row's maximum over the workgroup's 64 gate columns (the f16 scale) is the same kind of butterfly (the 16 owners of a row
//      sit in one wave), so the cell owner scales and splits ITS 4 values and the LDS image holds ready f16 operand fragments: the
//      MFMA phase starts with 4 ds_read_b128 instead of 16 conversions per lane in every wave.
// dynamic LDS: img[2 parities][hi | lo][16][GLD] f16 | rowexp[2][16] | wmax[8] | abort
// ================================================================================================
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
// x[l] combined with x[l ^ 16] / x[l ^ 32] (gfx950 v_permlane16_swap / v_permlane32_swap: no LDS crossbar)
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
__device__ __forceinline__ unsigned xmax16(unsigned u) {
  unsigned r[2] = {u, u};
  permlane16_swap(r[0], r[1]);
  return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xmax32(unsigned u) {
  unsigned r[2] = {u, u};
  permlane32_swap(r[0], r[1]);
  return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ float radd(float x, unsigned rot) { return x + __builtin_bit_cast(float, rot); }

template <int NMB, int BQ, int CELL, int NWV = 4>
__global__ void __launch_bounds__(64 * NWV) lstm_bwd5f_kernel(const LstmK p) {
  constexpr int NGATE = CELL == 0 ? 4 : (CELL == 1 ? 3 : 1);
  constexpr int HS = 16, NT = 64 * NWV, KSB = 2, NCMAX = 32;
  static_assert(NWV == 4 || NWV == 8, "waves 0..3 gather and own the cells; waves 4..7 (if any) only share the MFMA + publication phase");
  constexpr int NBR = 4 * BQ;        // exchange rows of the group
  constexpr int RPW = BQ;            // rows per gathering wave
  constexpr int PPW = 4 * RPW;       // (row, unit quad) pairs per wave: 4, 8, 16
  constexpr int NQ = 64 / PPW;       // producer classes = lanes per pair: 16, 8, 4
  constexpr int NLD = NCMAX / NQ;    // granules a lane sums in registers: 2, 4, 8
  constexpr int GLD = 32 * KSB + 8;  // f16 per image row (16-byte aligned rows)
  constexpr int OOB = 0x7ffffff0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* img = reinterpret_cast<_Float16*>(smem);                 // [2][2][16][GLD]
  unsigned* rowexp = reinterpret_cast<unsigned*>(img + 2 * 2 * 16 * GLD);
  float* wmax = reinterpret_cast<float*>(rowexp + 32);
  int* abort_lds = reinterpret_cast<int*>(wmax + 8);

  const int H = p.H, B = p.B, D = p.D, Kp = p.Kp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NG = D * p.G;
  const int gid = blockIdx.x % p.NGL, wg = blockIdx.x / p.NGL;
  if (gid >= NG) return;
  const int d = gid / p.G, g = gid % p.G;
  const int b0 = g * p.Bg, j0 = wg * HS;
  const int lrow = lane & 15, lq = lane >> 4;
  const int TT = p.T;                       // padded frames: strides of the stash
  const int T = group_steps(p, b0);         // steps this sync group runs (ragged batches: the longest of its rows)

  // W_hh slice as the A operand (as lstm_bwd5_kernel): lane -> output unit u = 16*(wave*NMB + mb) + lrow,
  // k = 32*ks + 8*lq + e = own gate column 4*unit + gate  ->  W_hh[gate*H + j0 + (k>>2)][u]
  f16x8 whi[NMB][KSB], wlo[NMB][KSB];
  float w_inv;
  {
    const float* W = p.w_hh[d];
    f32x4 raw[NMB][KSB][2];
    float m = 0.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int u = 16 * (wave * NMB + mb) + lrow;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const int c = 32 * ks + 8 * lq;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (u < H) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (e < NGATE) {
              lo[e] = W[(long)(e * H + j0 + (c >> 2)) * H + u];
              hi[e] = W[(long)(e * H + j0 + (c >> 2) + 1) * H + u];
            }
          }
        }
        raw[mb][ks][0] = lo;
        raw[mb][ks][1] = hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) wmax[wave] = m;
    if (tid == 0) *abort_lds = 0;
    if (tid < 32) rowexp[tid] = 0u;
    for (int i = tid; i < 2 * 2 * 16 * GLD / 2; i += NT) reinterpret_cast<unsigned*>(img)[i] = 0u;
    __syncthreads();
    m = wmax[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmaxf(m, wmax[w]);
    int eb = (int)((__float_as_uint(m) >> 23) & 255u);
    eb = eb < 15 ? 15 : eb;
    const float wscale = m > 0.f ? __uint_as_float((unsigned)(268 - eb) << 23) : 1.f;
    w_inv = m > 0.f ? __uint_as_float((unsigned)(eb - 14) << 23) : 1.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) split8h(raw[mb][ks][0], raw[mb][ks][1], wscale, whi[mb][ks], wlo[mb][ks]);
  }

  const long px_bytes = (long)p.NC * NBR * Kp * 4;  // one (parity, group) image: [producer][Kp / 16 blocks][4 * NBR granules]
  __amdgpu_buffer_rsrc_t px_rsrc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    px_rsrc[par] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.hx) + ((long)par * NG + gid) * px_bytes, 0, (int)px_bytes, RSRC_FLAGS);

  // lane = cls * PPW + uq * RPW + rr: exchange row wave*RPW + rr, unit quad uq, producer class cls; classes 0..3 own the cells
  const int rr = lane % RPW, ouq = (lane / RPW) & 3, cls = lane / PPW;
  const bool gw = wave < 4;          // gathering / cell-owning wave (scalar)
  const bool owner = gw && cls < 4;
  const int oi = cls & 3;
  const int brow = wave * RPW + rr, ob = b0 + brow, oj = j0 + 4 * ouq + oi;
  const bool valid = owner && brow < p.Bg && ob < B;
  const int olen = valid ? p.lens[ob] : 0;
  float dc_carry = 0.f;
  f32x4 db_acc = {0.f, 0.f, 0.f, 0.f}, dbh_acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 cmx = {0.f, 0.f, 0.f, 0.f}, cmxh = {0.f, 0.f, 0.f, 0.f};
  const int t_first = (d == 0) ? T - 1 : 0;
  const int tdir = (d == 0) ? -1 : 1;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)((long)TT * B * D * 4 * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, CELL == 1 ? (int)((long)TT * B * D * 4 * H * 4) : 0, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.cst, 0, CELL == 0 ? (int)((long)D * TT * B * H * 4) : 0, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long)TT * B * D * H * 4), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((long)TT * B * D * H * 4), RSRC_FLAGS);
  const bool addressed = valid;
  int g_off = addressed ? (int)(((((long)t_first * B + ob) * D + d) * 4 * H + 4 * oj) * 4) : OOB;
  int c_off = addressed ? (int)((((((long)d * TT + t_first) * (H / 4) + (oj >> 2)) * B + ob) * 4 + (oj & 3)) * 4) : OOB;
  int y_off = addressed ? (int)(((((long)t_first * B + ob) * D + d) * H + oj) * 4) : OOB;
  const int g_step = addressed ? tdir * B * D * 4 * H * 4 : 0, c_step = addressed ? tdir * H * B * 4 : 0, y_step = addressed ? tdir * B * D * H * 4 : 0;

  // exchange image of one producer: [unit / 16][row / RPW][unit quad][row % RPW] granules of 16 bytes (4 units of one row)
  const int gat_base = ((cls * (Kp >> 2) + (j0 >> 2)) * NBR + wave * PPW + (lane % PPW)) * 16;
  const int gat_step = NQ * NBR * Kp * 4;
  const int pub_base = lrow < NBR ? ((wg * (Kp >> 2) + 4 * wave * NMB) * NBR + ((lrow / RPW) * 4 + lq) * RPW + lrow % RPW) * 16 : OOB;
  constexpr int pub_mb = 4 * NBR * 16;
  // this workgroup consumes block wg of every producer: wave wg / NMB publishes it in its MFMA group (wg % NMB) / GB
  const int my_pause = ((wg % NMB) < ((NMB % 4 == 0 && NMB > 4) ? 4 : 2)) ? (p.pause & 255) : ((p.pause >> 8) & 255);
  const int img_w = brow * GLD + 4 * (4 * ouq + oi);   // the owner's 4 gate columns (f16 index inside one [16][GLD] plane)
  const int img_r = lrow * GLD + 8 * lq;
  unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = clock64();
  const bool local = p.allow_local && group_is_xcd_local(p.xcc + gid * p.NC, p.NC, wg, p.status, abort_lds);

  auto run = [&](auto local_tag) -> bool {
  constexpr bool LOCAL = decltype(local_tag)::value;
  u32x4 gr[NLD];
  auto issue_gather = [&](int s_next) {   // partials published at step s_next - 1
    const __amdgpu_buffer_rsrc_t src = px_rsrc[(s_next - 1) & 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool okp = gw && cls + NQ * i < p.NC;
      gr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, okp ? gat_base + i * gat_step : OOB, 0, AUX_SC1));
    }
  };
  auto tags_ok = [&](unsigned want) -> bool {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (gw && cls + NQ * i < p.NC) bad |= (gr[i][0] ^ want) | (gr[i][2] ^ want);
    return (bad & 1u) == 0u;
  };
  // the stash of a step (activated gates, c_t, c_{t-1}, dy) is fetched TWO steps ahead into the register set of the step's parity
  // (the loop body is instantiated twice, so no register is moved while its load is in flight): the reads come from HBM (the
  // stash of a layer is 0.8 GB at c2) and one step of 2 us does not cover that latency behind the poll traffic
  struct Stash { f32x4 gt; float c_t, c_p, dyv; };
  auto load_stash = [&](Stash& st, int ahead, int s_of) {   // the stash of step s_of = (current step) + ahead
    const bool in = s_of < T;
    st.gt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, in ? g_off + ahead * g_step : OOB, 0, 0));
    st.c_t = st.c_p = 0.f;
    if constexpr (CELL == 0) {
      st.c_t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, in ? c_off + ahead * c_step : OOB, 0, 0));
      st.c_p = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, s_of + 1 < T ? c_off + (ahead + 1) * c_step : OOB, 0, 0));
    } else if constexpr (CELL == 1) {
      st.c_p = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rsrc, s_of + 1 < T ? y_off + (ahead + 1) * y_step : OOB, 0, 0));
    }
    st.dyv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, in ? y_off + ahead * y_step : OOB, 0, 0));
  };
  auto drop = [&](float dyv, int ahead) -> float {   // the dropout mask of the layer output, regenerated from (seed, element index)
    return (hash_u32(p.seed, (unsigned long long)(unsigned)((y_off + ahead * y_step) >> 2)) >= p.drop_thresh) ? dyv * p.keep_scale : 0.f;
  };
  // (Tried: waves 4..7 of the 8-wave form fetching the stash for waves 0..3 through an LDS ring, so that no polling wave has an
  //  HBM-latency load in its in-order return queue: 8.76 instead of 7.95 ms of backward recurrences per c2 step.)
  Stash stA, stB;
  load_stash(stA, 0, 0);
  load_stash(stB, 1, 1);
  i32x4 st_dg = {0, 0, 0, 0}, st_dgh = {0, 0, 0, 0};
  int st_goff = OOB, st_t = 0;
  unsigned st_rm = 0u;
  auto flush_stash = [&]() {
    __builtin_amdgcn_raw_buffer_store_b128(st_dg, g_rsrc, st_goff, 0, 0);
    if constexpr (CELL == 1) __builtin_amdgcn_raw_buffer_store_b128(st_dgh, a_rsrc, st_goff, 0, 0);
    if constexpr (CELL != 1) {
      if (p.rowmax && st_rm != 0u) atomicMax(p.rowmax + (long)st_t * B + b0 + tid, st_rm);
    }
  };
  auto step = [&](const int s, Stash& cur) -> bool {
    f32x4& gt = cur.gt;
    float &c_t = cur.c_t, &c_p = cur.c_p, &dyv = cur.dyv;
    const int t = (d == 0) ? T - 1 - s : s;
    const int par = s & 1;
    const bool active = valid && t < olen;
    if (p.ydrop) dyv = drop(dyv, 0);
    DBG_STAMP(0);
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (s > 0) {
      const unsigned want = (((unsigned)(s - 1) >> 1) & 1u) ^ 1u;
      if (!__all(tags_ok(want))) ok = poll_tagged([&]() -> bool { issue_gather(s); return tags_ok(want); }, p.status);
      DBG_STAMP(1);
      // (Tried: the next step's stash loads right here, one step ahead into the other register set — as far ahead of the next gather as a
      //  load can be issued — instead of two steps ahead from inside the MFMA phase: c2 equal, c3 13.1 -> 14.0 ms.)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        u32x4 v = gr[i];
        v[0] &= ~1u;
        v[2] &= ~1u;
        gsum += __builtin_bit_cast(f32x4, v);   // producers beyond NC loaded zeros
      }
      // sum over the producer classes: lane bits log2(PPW) .. 5
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = gsum[e];
        if constexpr (PPW == 4) x = radd(x, dpp_u32<0x124>(__builtin_bit_cast(unsigned, x)));   // row_ror:4
        if constexpr (PPW <= 8) x = radd(x, dpp_u32<0x128>(__builtin_bit_cast(unsigned, x)));   // row_ror:8
        x = xadd16(x);
        x = xadd32(x);
        gsum[e] = x;
      }
    }
    auto stash_traffic = [&]() {   // this step's registers are free (the cell math is done): the stash of step s + 2, and the previous step's stores
      load_stash(cur, 2, s + 2);
      flush_stash();
    };
    DBG_STAMP(2);  // partial sums (registers + lane butterfly)
    f32x4 dg4 = {0.f, 0.f, 0.f, 0.f}, dgh4 = {0.f, 0.f, 0.f, 0.f};
    {
      float dh = (active ? dyv : 0.f) + (oi == 0 ? gsum[0] : (oi == 1 ? gsum[1] : (oi == 2 ? gsum[2] : gsum[3])));
      if (active) {
        if constexpr (CELL == 0) {
          const float ig = gt[0], fg = gt[1], gg = gt[2], og = gt[3];
          const float tc = tanh_hw(c_t);
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
          dg4 = (f32x4){dh * (1.f - hv * hv), 0.f, 0.f, 0.f};
          dgh4 = dg4;
        }
      } else {
        dc_carry = 0.f;
      }
      db_acc += dg4;
      if constexpr (CELL == 1) dbh_acc += dgh4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cmx[e] = fmaxf(cmx[e], fabsf(dg4[e]));
        if constexpr (CELL == 1) cmxh[e] = fmaxf(cmxh[e], fabsf(dgh4[e]));
      }
      // the row's maximum over the 64 own gate columns: its 16 owners are lanes of this wave (bits log2(RPW) .. log2(RPW)+3)
      unsigned mxb = umax(umax(__float_as_uint(dgh4[0]) & 0x7fffffffu, __float_as_uint(dgh4[1]) & 0x7fffffffu),
                          umax(__float_as_uint(dgh4[2]) & 0x7fffffffu, __float_as_uint(dgh4[3]) & 0x7fffffffu));
      if constexpr (RPW == 1) mxb = umax(mxb, dpp_u32<0xB1>(mxb));    // quad_perm [1,0,3,2]
      if constexpr (RPW <= 2) mxb = umax(mxb, dpp_u32<0x4E>(mxb));    // quad_perm [2,3,0,1]
      mxb = umax(mxb, dpp_u32<0x124>(mxb));                          // row_ror:4, row_ror:8 -> lanes l, l+4, l+8, l+12
      mxb = umax(mxb, dpp_u32<0x128>(mxb));
      if constexpr (RPW >= 2) mxb = xmax16(mxb);
      if constexpr (RPW == 4) mxb = xmax32(mxb);
      int eb = (int)((mxb >> 23) & 255u);
      eb = eb < 15 ? 15 : eb;
      const float gscale = mxb != 0u ? __uint_as_float((unsigned)(268 - eb) << 23) : 1.f;
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      f16x4 h4, l4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = dgh4[e] * gscale;
        const _Float16 h = (_Float16)v;
        h4[e] = h;
        l4[e] = (_Float16)(v - (float)h);
      }
      if (owner) {
        _Float16* dst = img + par * (2 * 16 * GLD) + img_w;
        *reinterpret_cast<f16x4*>(dst) = h4;
        *reinterpret_cast<f16x4*>(dst + 16 * GLD) = l4;
        if (lane < RPW) rowexp[par * 16 + brow] = mxb;
      }
    }
    if (!ok) *abort_lds = 1;
    __syncthreads();
    if (*abort_lds != 0) return false;
    DBG_STAMP(3);  // cell math + f16 image + barrier
    {
      const unsigned rmax = rowexp[par * 16 + lrow];
      int eb = (int)((rmax >> 23) & 255u);
      eb = eb < 15 ? 15 : eb;
      const float ginv = (rmax != 0u ? __uint_as_float((unsigned)(eb - 14) << 23) : 1.f) * w_inv;
      f16x8 ghi[KSB], glo[KSB];
      const _Float16* src = img + par * (2 * 16 * GLD) + img_r;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        ghi[ks] = *reinterpret_cast<const f16x8*>(src + 32 * ks);
        glo[ks] = *reinterpret_cast<const f16x8*>(src + 16 * GLD + 32 * ks);
      }
      const unsigned tag = (((unsigned)s >> 1) & 1u) ^ 1u;
      constexpr int GB = (NMB % 4 == 0 && NMB > 4) ? 4 : 2;
#pragma unroll
      for (int mb0 = 0; mb0 < NMB; mb0 += GB) {
        f32x4 acc[GB];
#pragma unroll
        for (int j = 0; j < GB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[mb0 + j][ks], ghi[ks], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb0 + j][ks], glo[ks], acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < GB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[mb0 + j][ks], ghi[ks], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
          u32x4 out = __builtin_bit_cast(u32x4, acc[j] * ginv);
          out[0] = (out[0] & ~1u) | tag;
          out[2] = (out[2] & ~1u) | tag;
          if constexpr (LOCAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, out), px_rsrc[s & 1], pub_base + pub_mb * (mb0 + j), 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, out), px_rsrc[s & 1], pub_base + pub_mb * (mb0 + j), 0, AUX_SC1);
        }
        if (mb0 == 0 && gw) stash_traffic();   // behind the first group's MFMAs and publication stores (as lstm_bwd5_kernel)
      }
    }
    DBG_STAMP(4);  // MFMA + publication
    if (gw) {
      for (int i = 0; i < my_pause; ++i) __builtin_amdgcn_s_sleep(1);
      issue_gather(s + 1);
    }
    st_dg = __builtin_bit_cast(i32x4, dg4);
    if constexpr (CELL == 1) st_dgh = __builtin_bit_cast(i32x4, dgh4);
    st_goff = g_off;
    st_t = t;
    if constexpr (CELL != 1) {
      st_rm = 0u;
      if (p.rowmax && tid < NBR && tid < p.Bg && b0 + tid < B) st_rm = rowexp[par * 16 + tid];   // (rewritten two steps later)
    }
    g_off += g_step;
    c_off += c_step;
    y_off += y_step;
    DBG_STAMP(5);  // gather issue + prefetch + stash
    return true;
  };
  for (int s = 0; s < T; s += 2) {
    if (!step(s, stA)) return false;
    if (s + 1 < T) {
      if (!step(s + 1, stB)) return false;
    }
  }
  flush_stash();
  return true;
  };
  const bool okrun = local ? run(std::true_type{}) : run(std::false_type{});
  if (!okrun) return;
  if (owner) {
    const long row = (long)gid * NBR + brow;
    *reinterpret_cast<f32x4*>(p.dbp + row * 4 * H + 4 * oj) = db_acc;
    if constexpr (CELL == 1) *reinterpret_cast<f32x4*>(p.dbhp + row * 4 * H + 4 * oj) = dbh_acc;
    if (p.colmax) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned a = __float_as_uint(cmx[e]);
        if (a != 0u) atomicMax(p.colmax + (long)d * 4 * H + 4 * oj + e, a);
        if constexpr (CELL == 1) {
          const unsigned b = __float_as_uint(cmxh[e]);
          if (b != 0u) atomicMax(p.colmax_h + (long)d * 4 * H + 4 * oj + e, b);
        }
      }
    }
  }
  if (p.dbg && tid == 0) {
    for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 8 + i] = dsum[i];
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    p.dbg[blockIdx.x * 8 + 6] = local ? 1 : 0;
    p.dbg[blockIdx.x * 8 + 7] = xcc_id & 0xf;
  }
}

}  // namespace

// host side ----------------------------------------------------------------------------------------------------
// v5 takes H in {128, 256, 384, 512} (4 waves x 16 units) and H = 640 (8 waves x 20 units, K padded to 768), cells LSTM / GRU /
// tanh RNN; everything else stays on v3 / v4
bool lstm5_supported(int T, int B, int H, int D, int cell) {
  if (getenv("RNNT_LSTM_NO_V5") || getenv("RNNT_LSTM_V1") || getenv("RNNT_LSTM_V2") || getenv("RNNT_LSTM_EXACT_MATH")) return false;
  if ((long)T * B * D * 4 * H * 4 >= (1l << 31)) return false;   // the stash is addressed with 32-bit buffer offsets
  const bool h640 = H == 640 && !getenv("RNNT_LSTM_NO_H640_FORM") && !getenv("RNNT_LSTM_NO_8WAVE");
  // H = 768 / 1024 (8 waves, 48 / 64 workgroups per group): opt-in.  Such a group spans XCDs, every step moves 16 rows x H x 4 B to
  // each of its workgroups through the fabric, and the shipped config (8 x 1024 bi-GRU, B = 16) measured 152.5 ms per step against
  // 146.7 with lstm.hip's v3 / v4 forms (154.0 vs 145.6 with one 16-row group per direction instead of two 8-row groups).
  const bool wide = (H == 768 || H == 1024) && getenv("RNNT_LSTM_V5_WIDE") && !getenv("RNNT_LSTM_NO_8WAVE");
  return ((H % 128 == 0 && H >= 128 && H <= 512) || h640 || wide) && cell != RNNT_CELL_RNN_RELU;
}

static int env_pause(const char* name) {   // "e" or "e,l"
  const char* v = getenv(name);
  if (!v) return -1;
  int e = atoi(v), l = e;
  if (const char* c = strchr(v, ',')) l = atoi(c + 1);
  return (e & 255) | ((l & 255) << 8);
}

int lstm5_fwd_launch(const LstmK& k_in, const Plan2& pl, int cell, hipStream_t s) {
  LstmK k = k_in;
  // A pause between a step's publication and its first poll round: the round issued right behind the publication never finds the
  // operands (they are one L2 hand-off away) and its requests queue in front of the round that would.  Measured (paired bench runs,
  // profiles/r02_poll_pause_ab.txt): 8-row groups at H = 512 (c2) 6.30 -> 6.10 ms of forward recurrences per step with 8 x 64 cycles,
  // H = 640 (c5) 17.8 -> 17.4 with 6; 4-row groups (c3) lose 3 % with any pause.
  { const int e = env_pause("RNNT_LSTM_FWD_PAUSE"); k.pause = e >= 0 ? e : (pl.Bg >= 8 ? (k.H == 640 ? 6 : 8) : 0); }
  const int nks = k.Kp / 128;
  int rc = RNNT_ERR_UNSUPPORTED;
  if (pl.MB == 5) {   // H = 640
    const size_t lds = (size_t)2 * 8 * 5 * 64 * 16 + 32 + 16 + 16 * 20 * 4;
    if (cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd5_kernel<3, 0, 8, 5>, k, pl, lds, s, "lstm_fwd5", 512);
    else if (cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd5_kernel<3, 1, 8, 5>, k, pl, lds, s, "lstm_fwd5", 512);
    else rc = launch_persistent2(lstm_fwd5_kernel<3, 2, 8, 5>, k, pl, lds, s, "lstm_fwd5", 512);
    return rc;
  }
  // H = 512: 8 waves x 2 k-steps instead of 4 x 4.  The MFMA phase is pipe-bound either way (192 MFMAs per workgroup and step on 4
  // SIMDs = 768 cycles) but 8 waves spread the operand polls, the de-interleave and the stash over twice the issue slots:
  // 4 900 vs 5 324 cycles per step, 31.4 vs 32.1 ms per c2 step (profiles/r02_lstm_phase_cycles_v5.txt).  RNNT_LSTM_FWD5_4W=1: 4 waves.
  const bool w8 = k.H == 512 && !getenv("RNNT_LSTM_FWD5_4W") && !getenv("RNNT_LSTM_NO_8WAVE");
  if (k.H > 512 || w8) {   // H = 768 / 1024: 8 waves x 3 / 4 k-steps each
    const size_t lds = (size_t)2 * 8 * 4 * 64 * 16 + 32 + 16 + 16 * 16 * 4;
#define L58(N)                                                                                                       \
    do {                                                                                                             \
      if (cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd5_kernel<N, 0, 8, 4>, k, pl, lds, s, "lstm_fwd5", 512);      \
      else if (cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd5_kernel<N, 1, 8, 4>, k, pl, lds, s, "lstm_fwd5", 512);  \
      else rc = launch_persistent2(lstm_fwd5_kernel<N, 2, 8, 4>, k, pl, lds, s, "lstm_fwd5", 512);                   \
    } while (0)
    if (k.Kp == 768) L58(3);
    else if (k.Kp == 1024) L58(4);
    else if (k.Kp == 512) L58(2);
    else set_error("lstm_fwd5: H = %d not supported", k.H);
#undef L58
    return rc;
  }
  const size_t lds = (size_t)2 * 4 * 4 * 64 * 16 + 32 + 16 + 16 * 16 * 4;
#define L5(N)                                                                                              \
  do {                                                                                                     \
    if (cell == RNNT_CELL_LSTM) rc = launch_persistent2(lstm_fwd5_kernel<N, 0>, k, pl, lds, s, "lstm_fwd5");      \
    else if (cell == RNNT_CELL_GRU) rc = launch_persistent2(lstm_fwd5_kernel<N, 1>, k, pl, lds, s, "lstm_fwd5");  \
    else rc = launch_persistent2(lstm_fwd5_kernel<N, 2>, k, pl, lds, s, "lstm_fwd5");                      \
  } while (0)
  if (nks == 1) L5(1);
  else if (nks == 2) L5(2);
  else if (nks == 3) L5(3);
  else if (nks == 4) L5(4);
  else set_error("lstm_fwd5: H = %d not supported", k.H);
#undef L5
  return rc;
}

int lstm5_bwd_launch(const LstmK& k_in, const Plan2& pl, int cell, hipStream_t s) {
  LstmK k = k_in;
  // one-barrier form: c2 7.83 -> 7.55 ms of backward recurrences per step with 8 x 64 cycles, c3 13.4 -> 13.1; 12 and more lose again
  const int env_p = env_pause("RNNT_LSTM_BWD_PAUSE");
  k.pause = env_p >= 0 ? env_p : 0;   // lstm_bwd5_kernel (H = 640 at c5): any pause loses (22.3 -> 22.6 ms with 4, 24.0 with 12)
  const int nks = k.Kp / 128;
  int rc = RNNT_ERR_UNSUPPORTED;
#define B5Q(NM, C, ...)                                                                               \
  do {                                                                                                \
    if (pl.BQ == 1) rc = launch_persistent2(lstm_bwd5_kernel<NM, 1, C, ##__VA_ARGS__>, k, pl, lds, s, "lstm_bwd5", threads);       \
    else if (pl.BQ == 2) rc = launch_persistent2(lstm_bwd5_kernel<NM, 2, C, ##__VA_ARGS__>, k, pl, lds, s, "lstm_bwd5", threads);  \
    else rc = launch_persistent2(lstm_bwd5_kernel<NM, 4, C, ##__VA_ARGS__>, k, pl, lds, s, "lstm_bwd5", threads);                  \
  } while (0)
  if (pl.MB == 5) {   // H = 640: 80 own gate columns (3 k-steps), 48 output blocks over 8 waves
    const int threads = 512;
    const size_t lds = (size_t)512 * 16 + 16 * (32 * 3 + 4) * 4 + 32 * 4 + 32 + 16;
    if (cell == RNNT_CELL_LSTM) B5Q(6, 0, 8, 5);
    else if (cell == RNNT_CELL_GRU) B5Q(6, 1, 8, 5);
    else B5Q(6, 2, 8, 5);
    return rc;
  }
  // (H = 512 with 8 waves x 4 output blocks: the MFMA + publication phase drops from 1 961 to 1 432 cycles but the wait for the
  //  partial sums grows from 768 to 1 894: 5 953 vs 5 156 cycles per step -> stays at 4 waves)
  if (k.H > 512) {   // H = 768 / 1024: 8 waves, 6 / 8 output blocks each, up to 64 producers per group
    const int threads = 512;
    const size_t lds = (size_t)512 * 16 + 16 * (32 * 2 + 4) * 4 + 32 * 4 + 32 + 16;
#define B58(NM)                                          \
    do {                                                 \
      if (cell == RNNT_CELL_LSTM) B5Q(NM, 0, 8, 4, 64);  \
      else if (cell == RNNT_CELL_GRU) B5Q(NM, 1, 8, 4, 64); \
      else B5Q(NM, 2, 8, 4, 64);                         \
    } while (0)
    if (k.Kp == 768) B58(6);
    else if (k.Kp == 1024) B58(8);
    else set_error("lstm_bwd5: H = %d not supported", k.H);
#undef B58
    return rc;
  }
  if (k.H == 512 && getenv("RNNT_LSTM_BWD5_8W")) {   // 8 waves x 4 output blocks (two waves per SIMD share the MFMA pipe)
    const int threads = 512;
    const size_t lds = (size_t)512 * 16 + 16 * (32 * 2 + 4) * 4 + 32 * 4 + 32 + 16;
    if (cell == RNNT_CELL_LSTM) B5Q(4, 0, 8, 4, 32);
    else if (cell == RNNT_CELL_GRU) B5Q(4, 1, 8, 4, 32);
    else B5Q(4, 2, 8, 4, 32);
    return rc;
  }
  if (!getenv("RNNT_LSTM_BWD5_2B")) {   // one-barrier form (default); RNNT_LSTM_BWD5_2B=1: lstm_bwd5_kernel (A/B partner)
    if (env_p < 0) k.pause = 8 | (8 << 8);
    const int threads = 256;
    const size_t lds = (size_t)2 * 2 * 16 * (32 * 2 + 8) * 2 + 32 * 4 + 32 + 16;
#define B5F(NM)                                                                                                   \
    do {                                                                                                          \
      if (cell == RNNT_CELL_LSTM) { if (pl.BQ == 1) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 1, 0>, k, pl, lds, s, "lstm_bwd5f", threads);       \
        else if (pl.BQ == 2) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 2, 0>, k, pl, lds, s, "lstm_bwd5f", threads);                             \
        else rc = launch_persistent2(lstm_bwd5f_kernel<NM, 4, 0>, k, pl, lds, s, "lstm_bwd5f", threads); }                                             \
      else if (cell == RNNT_CELL_GRU) { if (pl.BQ == 1) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 1, 1>, k, pl, lds, s, "lstm_bwd5f", threads);  \
        else if (pl.BQ == 2) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 2, 1>, k, pl, lds, s, "lstm_bwd5f", threads);                             \
        else rc = launch_persistent2(lstm_bwd5f_kernel<NM, 4, 1>, k, pl, lds, s, "lstm_bwd5f", threads); }                                             \
      else { if (pl.BQ == 1) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 1, 2>, k, pl, lds, s, "lstm_bwd5f", threads);                             \
        else if (pl.BQ == 2) rc = launch_persistent2(lstm_bwd5f_kernel<NM, 2, 2>, k, pl, lds, s, "lstm_bwd5f", threads);                             \
        else rc = launch_persistent2(lstm_bwd5f_kernel<NM, 4, 2>, k, pl, lds, s, "lstm_bwd5f", threads); }                                             \
    } while (0)
    if (nks == 4 && !getenv("RNNT_LSTM_BWD5F_4W")) {   // H = 512: 8 waves x 4 output blocks (two waves per SIMD keep the MFMA pipe busy; one wave issues a
      const int threads8 = 512;                         // v_mfma_f32_16x16x32_f16 only every 20-25 cycles), waves 4..7 share the MFMA + publication phase only
#define B5F8(BQV, C) rc = launch_persistent2(lstm_bwd5f_kernel<4, BQV, C, 8>, k, pl, lds, s, "lstm_bwd5f", threads8)
      if (cell == RNNT_CELL_LSTM) { if (pl.BQ == 1) B5F8(1, 0); else if (pl.BQ == 2) B5F8(2, 0); else B5F8(4, 0); }
      else if (cell == RNNT_CELL_GRU) { if (pl.BQ == 1) B5F8(1, 1); else if (pl.BQ == 2) B5F8(2, 1); else B5F8(4, 1); }
      else { if (pl.BQ == 1) B5F8(1, 2); else if (pl.BQ == 2) B5F8(2, 2); else B5F8(4, 2); }
#undef B5F8
      return rc;
    }
    if (nks == 1) B5F(2);
    else if (nks == 2) B5F(4);
    else if (nks == 3) B5F(6);
    else if (nks == 4) B5F(8);
    else set_error("lstm_bwd5: H = %d not supported", k.H);
#undef B5F
    return rc;
  }
  const int threads = 256;
  const size_t lds = (size_t)256 * 16 + 16 * (32 * 2 + 4) * 4 + 32 * 4 + 32 + 16;
#define B5(NM)                                       \
  do {                                               \
    if (cell == RNNT_CELL_LSTM) B5Q(NM, 0);          \
    else if (cell == RNNT_CELL_GRU) B5Q(NM, 1);      \
    else B5Q(NM, 2);                                 \
  } while (0)
  if (nks == 1) B5(2);
  else if (nks == 2) B5(4);
  else if (nks == 3) B5(6);
  else if (nks == 4) B5(8);
  else set_error("lstm_bwd5: H = %d not supported", k.H);
#undef B5
#undef B5Q
  return rc;
}

}  // namespace rnnt
