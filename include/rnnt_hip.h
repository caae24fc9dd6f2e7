/*
 * rnnt_hip.h — C ABI of librnnt_hip.so: the MI355X (gfx950) native RNN-Transducer training hot path.
 *
 * The reference (YooSungHyun/RNNTransducer) has NO native code and therefore no FFI of its own
 * (SURVEY.md §0, §2b): every kernel it runs comes from a dependency.  This header declares the entry
 * points a maintainer would bind in place of those dependency calls; each one cites the reference call
 * site it replaces (paths relative to the reference repo).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions:
 *   - plain `extern "C"`, raw device pointers + explicit dims, `void* stream` is a hipStream_t;
 *   - return 0 on success, <0 on error; rnnt_hip_last_error() gives a thread-local message;
 *   - the CALLER owns all memory incl. workspaces (sizes from *_workspace_bytes); the library never
 *     allocates or frees device memory, never synchronises the device (the two *_check / *_debug_read
 *     diagnostics aside) and keeps no global mutable state other than the opt-in profiler below
 *     (off by default: event lists + a mutex behind rnnt_hip_prof_enable) and the thread-local error string;
 *   - all float tensors are fp32, all lengths/labels int32, token ids int64 (dataloader.py:21-24,28-36);
 *   - "time-major" = (T,B,F) contiguous; "batch-major" = (B,T,F) contiguous.
 */
#ifndef RNNT_HIP_H_
#define RNNT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNNT_HIP_ABI_VERSION 4

#define RNNT_OK 0
#define RNNT_ERR_INVALID (-1)   /* bad argument (dims, alignment, null pointer)            */
#define RNNT_ERR_LAUNCH (-2)    /* hip launch / runtime error                              */
#define RNNT_ERR_UNSUPPORTED (-3) /* configuration outside what the kernels handle         */
#define RNNT_ERR_TIMEOUT (-4)   /* a persistent kernel gave up on an inter-CU wait         */

int rnnt_hip_version(void);
const char* rnnt_hip_last_error(void);
/* number of compute units the persistent LSTM kernels size their grid against (0 = no device) */
int rnnt_hip_device_cus(void);

/* ------------------------------------------------------------------------------------------------
 * Opt-in live profiler (used by bench.py only).  While enabled, every kernel launch below is bracketed by two
 * HIP events recorded on the launch stream; collect() synchronises on them, sums elapsed ms / algorithmic work /
 * launch counts per kernel kind, and resets.  This is the only global state in the library, off by default.
 * `work` unit: FLOPs for RNNT_K_GEMM and RNNT_K_GEMM_HP, algorithmic bytes for all other kinds.
 * ---------------------------------------------------------------------------------------------- */
enum {
  RNNT_K_GEMM = 0,       /* gemm_f32_kernel                                   */
  RNNT_K_LSTM_FWD = 1,   /* lstm_fwd_kernel (persistent recurrence)           */
  RNNT_K_LSTM_BWD = 2,   /* lstm_bwd_kernel                                   */
  RNNT_K_LSE = 3,        /* lse_sep_kernel / lse_dense_kernel                 */
  RNNT_K_ALPHABETA = 4,  /* alphabeta_kernel                                  */
  RNNT_K_LATGRAD = 5,    /* grad_sep_kernel / grad_dense_kernel + reduce_dc   */
  RNNT_K_MISC = 6,       /* permutes, column sums, embedding, logits          */
  RNNT_K_GEMM_HP = 7,    /* gemm_hp_kernel (half-pair operands, f16 MFMA)     */
  RNNT_K_HP_SPLIT = 8,   /* fp32 -> half-pair operand conversion passes       */
  RNNT_K_COUNT = 9
};
int rnnt_hip_prof_enable(int on);
int rnnt_hip_prof_collect(double* ms, double* work, int64_t* count, int nkinds);

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on f32-input MFMA (v_mfma_f32_32x32x2_f32):  C = op(A) . op(B) (+ bias)
 * Replaces the BLAS calls behind nn.Linear / the hoisted LSTM input projection:
 *   networks/encoder.py:76,103 (out_proj), networks/decoder.py:80,124 (out_proj),
 *   networks/transducer.py:39,69 (fc), and the W_ih.x_t half of nn.LSTM (encoder.py:67-75,99).
 *
 *   A(m,k) = A[rowoff_a(m) + k*a_sk]            if a_mc == 0   (k-contiguous rows, a_sk must be 1)
 *   A(m,k) = A[k*a_sk + m]                      if a_mc == 1   (m-contiguous, "transposed" operand)
 *     rowoff_a(m) = a_rowidx ? a_rowidx[m]*a_si : (m / a_div)*a_so + (m % a_div)*a_si
 *   B(k,n) = B[n*b_sn + k*b_sk]                 exactly one of b_sn, b_sk is 1
 *   C(m,n) = C[(m / c_div)*c_so + (m % c_div)*c_si + n]
 * flags: see RNNT_GEMM_*.
 *
 * Arithmetic: inputs, outputs and accumulators are fp32 in every mode.  Default (RNNT_GEMM_MODE unset or "bf16x6"):
 * each fp32 operand is split EXACTLY into three bf16 pieces and the six piece products of fp32 weight run on
 * v_mfma_f32_32x32x16_bf16 (error per product <= ~3 * 2^-24, the order of fp32 rounding itself).  "f32": the
 * f32-input MFMA (exact fp32 fma chain).  "bf16x3": first-order pieces only (~2^-16 per product), opt-in.
 * ---------------------------------------------------------------------------------------------- */
#define RNNT_GEMM_GELU_A 1u      /* apply gelu_tanh to A elements on load  (transducer.py:38,68)   */
#define RNNT_GEMM_GELU_B 2u      /* apply gelu_tanh to B elements on load                          */
#define RNNT_GEMM_ACCUM 4u       /* C += result                                                    */
#define RNNT_GEMM_MUL_DGELU 8u   /* C = result * gelu_tanh'(aux(m,n)), aux laid out like C          */
#define RNNT_GEMM_EXACT_F32 16u  /* multiply on v_mfma_f32_32x32x2_f32 (bit-exact fp32 fma chains) even when the
                                  * library default is the split-bf16 form (see below)                      */

typedef struct rnnt_gemm_desc {
  int64_t M, N, K;
  const float* A;
  int64_t a_div, a_so, a_si, a_sk;
  int32_t a_mc;
  const int64_t* a_rowidx;
  const float* B;
  int64_t b_sn, b_sk;
  float* C;
  int64_t c_div, c_so, c_si;
  const float* bias; /* (N) or NULL */
  const float* aux;  /* for RNNT_GEMM_MUL_DGELU */
  uint32_t flags;
  void* workspace;   /* optional: enables deterministic split-K (slabs + fixed-order reduce) for GEMMs whose   */
  size_t workspace_bytes; /* output has too few tiles to fill the chip; see rnnt_hip_gemm_workspace_bytes     */
} rnnt_gemm_desc;

size_t rnnt_hip_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int rnnt_hip_gemm_f32(const rnnt_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 GEMM on the f16 matrix cores through "half-pair" (hp) operands — the form the BIG products of the hot path use
 * inside rnnt_hip_lstm_fwd / _bwd (hoisted input projection of nn.LSTM, networks/encoder.py:67-75,99, and its backward
 * products dX, dW_ih, dW_hh): 3 instead of 6 MFMA products per fp32 product, operands streamed to LDS by LDS-DMA.
 *
 * hp planes of an fp32 matrix x (rows x K), scaled per row: amax[r] = max_k |x[r][k]| (fp32 bit patterns, `rows` device words
 *   next to the planes), v = x * 2^(14 - floor(log2 amax[r])), hi = fp16_rn(v), lo = fp16_rn(v - hi)
 *   (v = hi + lo to 2^-23 |v|, with a floor of 2^-40 amax[r]); layout: row-major, K padded to 32, one 128-byte line per
 *   (row, 32-k block): 32 hi | 32 lo.  rnnt_hip_hp_bytes(rows, K).
 * rnnt_hip_hp_split: transpose == 0: x is (rows x K), row stride ld; amax[rows] is written (amax_given must be 0).
 *   transpose == 1: x is (src_rows x >= rows), row stride ld; plane row r, index k holds x[k + shift][r] (0 outside
 *   [0, src_rows)) — the transposed operands of the weight-gradient products, time-shifted for dW_hh; amax[r] = maximum of source
 *   column r, computed here unless amax_given.
 * rnnt_hip_gemm_hp: C (M x N, row stride ldc) [+]= A (M x K) . B (N x K)^T + bias, both operands hp planes (NT form), fp32 out.
 *   flags: RNNT_GEMM_ACCUM.  workspace (optional, rnnt_hip_gemm_hp_workspace_bytes): deterministic split-K slabs.
 * ---------------------------------------------------------------------------------------------- */
size_t rnnt_hip_hp_bytes(int64_t rows, int64_t K);
int rnnt_hip_hp_split(const float* x, int64_t rows, int64_t K, int64_t ld, int32_t transpose, int64_t src_rows, int64_t shift,
                      void* planes, uint32_t* amax, int32_t amax_given, void* stream);
/* both orientations of x (M x C, row stride ld) in one pass: planes_rm = what transpose == 0 writes given the row maxima rowmax[M],
 * planes_t = what transpose == 1 (shift 0, contraction length M) writes given the column maxima colmax[C]; both tables are inputs
 * (rnnt_hip_lstm_bwd's recurrence leaves them for dG).  Bitwise the same planes as two rnnt_hip_hp_split calls. */
int rnnt_hip_hp_split_both(const float* x, int64_t M, int64_t C, int64_t ld, const uint32_t* rowmax, const uint32_t* colmax,
                           void* planes_rm, void* planes_t, void* stream);
size_t rnnt_hip_gemm_hp_workspace_bytes(int64_t M, int64_t N, int64_t K);
int rnnt_hip_gemm_hp(const void* A, const uint32_t* a_amax, const void* B, const uint32_t* b_amax, int64_t M, int64_t N, int64_t K,
                     float* C, int64_t ldc, const float* bias, uint32_t flags, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Up to 4 such products in ONE queue-driven launch: 256 resident workgroups draw (problem, tile, K-split) units until none is left
 * (one launch tail instead of one per product).  xcd_skip: bit x set = workgroups that find themselves on XCD x leave at once, the
 * other XCDs do all the work — for products that run on a second stream beside a persistent recurrence (rnnt_lstm_bwd_desc.phase).
 * workspace: rnnt_hip_gemm_hp_grouped_workspace_bytes(...) bytes, 256-byte aligned (queue counters + deterministic split-K slabs).
 * Self-check: a kernel behind the launch compares the units completed with the units queued; if they differ (xcd_skip named XCDs the
 * device does not expose, so every workgroup left) it sets word 9 of `workspace` to 1 — callers of this entry that pass a non-zero
 * xcd_skip read it back; rnnt_hip_lstm_bwd raises its sticky status word instead (and only passes a mask on a 256-CU device).
 * Used internally by rnnt_hip_lstm_bwd for dW_ih / dW_hh; exposed for tests. */
typedef struct rnnt_hp_problem {
  const void* A; const uint32_t* a_amax;   /* (M x K) planes + row maxima */
  const void* B; const uint32_t* b_amax;   /* (N x K) */
  int64_t M, N, K;
  float* C; int64_t ldc;
  uint32_t flags;                          /* RNNT_GEMM_ACCUM */
} rnnt_hp_problem;
size_t rnnt_hip_gemm_hp_grouped_workspace_bytes(const rnnt_hp_problem* problems, int32_t n);
int rnnt_hip_gemm_hp_grouped(const rnnt_hp_problem* problems, int32_t n, uint32_t xcd_skip, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSTM layer (both directions in one launch), packed-sequence semantics.
 * Replaces torch.nn.LSTM over a PackedSequence + sort/pack/unpack/unsort:
 *   networks/encoder.py:67-75 (ctor), :93-102 (forward);  networks/decoder.py:71-79, :105-120.
 *
 * Layouts: x (T,B,I) time-major [or any (x_st, x_sb) element strides], y (T,B,D*H) time-major.
 *   Frames t >= lens[b] produce y == 0 and contribute no gradient (pad_packed_sequence semantics,
 *   encoder.py:101); the reverse direction starts at each sequence's own last valid frame.
 *   Weights in torch layout: w_ih[d] (G*H,I), w_hh[d] (G*H,H), b_ih[d], b_hh[d] (G*H); G = 4 (LSTM: i,f,g,o),
 *   3 (GRU: r,z,n), 1 (Elman RNN).  The gate buffer always has 4 slots per hidden unit.
 *   Requirements: H % 4 == 0, D in {1,2}, B <= 64 per call, T >= 1.
 *
 * Stash written by fwd and consumed by bwd (caller-owned, sizes below):
 *   gates: D*T*B*4H floats ... activated gates, overwritten with dG by bwd
 *   cst  : D*T*B*H  floats ... cell states
 * ---------------------------------------------------------------------------------------------- */
#define RNNT_CELL_LSTM 0      /* gates i,f,g,o   weights (4H, .)                                        */
#define RNNT_CELL_GRU 1       /* gates r,z,n     weights (3H, .)  (torch.nn.GRU layout and equations)    */
#define RNNT_CELL_RNN_TANH 2  /* Elman           weights (H, .)                                          */
#define RNNT_CELL_RNN_RELU 3

typedef struct rnnt_lstm_desc {
  int32_t T, B, I, H, D;
  int32_t cell;        /* RNNT_CELL_*: the reference's supported_rnns = lstm | gru | rnn (encoder.py:48-52) */
  const int32_t* lens; /* (B) device */
  const float* x;
  int64_t x_st, x_sb; /* element strides of x over t and b (feature stride 1) */
  const float* w_ih[2];
  const float* w_hh[2];
  const float* b_ih[2];
  const float* b_hh[2];
  float* y;       /* (T,B,D*H) */
  float* y_drop;  /* (T,B,D*H) y with inter-layer dropout applied, or NULL when dropout_p == 0 */
  float dropout_p;
  uint64_t dropout_seed;
  float* gates;   /* (T,B,D*4H) permuted gate layout, see DESIGN.md */
  float* cst;     /* (D,T,H/4,B,4)  LSTM only (may be NULL for the other cells) */
  float* aux;     /* GRU backward only: (T,B,D*4H) scratch for the hidden-side gate gradients; else NULL */
  void* workspace;
  size_t workspace_bytes;
  uint32_t* status; /* optional: caller-owned STICKY device status word (4 bytes, zeroed once by the caller).  A persistent
                     * kernel that abandons an inter-workgroup wait (4 s bound; e.g. its workgroups lost co-residency to a
                     * concurrent kernel) stores 1 here; the library never clears it, every later launch handed the same
                     * word bails out at its first wait, and rnnt_hip_adamw_step_ex(guard = this word) skips the update.
                     * Read it back with an asynchronous 4-byte copy whenever convenient (rnntransducer_amd does so once per
                     * optimizer step, no extra synchronisation).  NULL: word 0 of the workspace, reset per launch, read by
                     * rnnt_hip_lstm_check(). */
  float x_abs_bound; /* optional (backward): > 0 = the caller guarantees |x| <= x_abs_bound everywhere (x is the dropped output of
                     * a bounded cell below: 1 / (1 - p)).  The half-pair planes of x^T then take this as their scale instead of
                     * a pass over x for its column maxima (absolute error of an element <= 2^-39 x_abs_bound either way).  0: measure. */
  const int32_t* row_idx; /* optional, ragged batches (what pack_padded_sequence buys the reference, encoder.py:93-96,99-101, without a packed
                     * copy): device table of the n_rows VALID time-major rows t*B + b (those with t < lens[b]), ascending.  The big products
                     * (input projection, dX, dW_ih, dW_hh) then run over n_rows instead of T*B rows — operand tiles are gathered / results
                     * scattered through this table — and every sync group of the recurrence runs max(lens of its rows) steps instead of T
                     * (reverse direction: from that frame down).  Rows that are not listed are then NOT written in gates / cst / y_drop,
                     * and y keeps what the caller put there: hand in y zero-filled (frames t >= lens[b] must read 0).  Results on valid
                     * frames do not depend on it.  Honoured by the default kernels (v5 recurrences + half-pair products); other shapes
                     * ignore it and compute all T*B rows.  NULL: all rows.  Same table for the forward and the backward call. */
  int32_t n_rows;   /* entries of row_idx (= sum of lens); ignored when row_idx is NULL */
} rnnt_lstm_desc;

size_t rnnt_hip_lstm_workspace_bytes(int32_t T, int32_t B, int32_t I, int32_t H, int32_t D);
/* largest B one call accepts for (H, D, cell); 0 = shape unsupported.  Bigger batches: split along B (rows are independent). */
int32_t rnnt_hip_lstm_max_batch(int32_t H, int32_t D, int32_t cell);
/* XCDs (of 8) a recurrence of this shape leaves without a workgroup (0 when its groups fill the chip or are not placed per XCD):
 * what a caller looks at before it puts phase 2 of one layer beside phase 1 of the next (rnnt_lstm_bwd_desc.phase). */
int32_t rnnt_hip_lstm_free_xcds(int32_t T, int32_t B, int32_t H, int32_t D, int32_t cell);
/* 1 if a layer of this shape honours rnnt_lstm_desc.row_idx (v5 recurrences + half-pair products: every consumer of the stash gathers
 * the valid rows), 0 if it ignores the table and computes all T*B rows. */
int32_t rnnt_hip_lstm_takes_row_idx(int32_t T, int32_t B, int32_t I, int32_t H, int32_t D, int32_t cell);
int rnnt_hip_lstm_fwd(const rnnt_lstm_desc* d, void* stream);

typedef struct rnnt_lstm_bwd_desc {
  rnnt_lstm_desc f;   /* same description as the forward call (x, weights, y, stash, workspace) */
  const float* dy;    /* (T,B,D*H) gradient w.r.t. y (w.r.t. y_drop when dropout_p > 0) */
  float* dx;          /* (T,B,I) time-major, or NULL (first layer: dataloader.py gives no grad to mel) */
  float* dw_ih[2];    /* (4H,I)  written (not accumulated) */
  float* dw_hh[2];    /* (4H,H) */
  float* db[2];       /* (G*H)  gradient of b_ih (== gradient of b_hh for LSTM / RNN) */
  float* db_hh[2];    /* (G*H)  GRU: gradient of b_hh (differs from b_ih in the n gate), required.  LSTM / RNN: optional second
                       * destination that receives the same values as db (grad b_hh == grad b_ih), or NULL */
  int32_t accumulate; /* 0: dw_ih / dw_hh / db / db_hh are written; 1: added to (the outputs are views of a flat gradient
                       * buffer that autograd would otherwise `+=` into with one extra kernel per parameter) */
  int32_t phase;      /* RNNT_LSTM_BWD_ALL (0): everything on `stream`.  The two halves can also be issued separately so that a
                       * caller overlaps the weight gradients of layer l with the recurrence of layer l-1 (autograd needs only dx
                       * to go on): RNNT_LSTM_BWD_RECUR (1) = reverse-time recurrence (gates -> dG in place) + dx;
                       * RNNT_LSTM_BWD_WEIGHTS (2) = dw_ih / dw_hh / db / db_hh from the dG that phase 1 left in `gates`, on any
                       * stream ordered after phase 1, with the SAME descriptor and workspace (which phase 1 of another layer must
                       * not reuse before phase 2 is done: alternate two workspaces). */
  int32_t beside_recurrence; /* phase 2 only, a hint: 1 = a recurrence of the same (B,H,D) runs concurrently on another stream; the
                       * big products then leave the XCDs that recurrence occupies alone (its workgroups exchange through their
                       * XCD's L2) and run as one queue-driven launch on the others.  Results do not depend on it (honoured only on a
                       * device that exposes all 8 XCDs; a launch that left work undone raises `f.status`). */
} rnnt_lstm_bwd_desc;
#define RNNT_LSTM_BWD_ALL 0
#define RNNT_LSTM_BWD_RECUR 1
#define RNNT_LSTM_BWD_WEIGHTS 2

int rnnt_hip_lstm_bwd(const rnnt_lstm_bwd_desc* d, void* stream);
/* reads back the persistent kernels' status word from a workspace (synchronises `stream`);
 * 0 = ok, RNNT_ERR_TIMEOUT if an inter-CU wait gave up.  For tests and the bench, not for hot loops. */
int rnnt_hip_lstm_check(const void* workspace, void* stream);
/* Diagnostics (a library built with -DRNNT_LSTM_DBG_STAMPS=1 only: the stamps are compiled out of the default build, where this
 * entry returns zeros): with RNNT_LSTM_DBG set in the environment the recurrences accumulate shader-clock cycles per step
 * phase (0 prefetch issue, 1 flag wait, 2 gather+MFMA, 3 reduce+cell math, 4 drain+barrier+flag, 5 stash stores) for
 * lane 0 of every workgroup; this copies the last launch's table (nwg x 8 u64) to host memory (synchronises). */
int rnnt_hip_lstm_debug_read(const void* workspace, int32_t T, int32_t B, int32_t I, int32_t H, int32_t D,
                             uint64_t* out, int32_t nwg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused joint + log-softmax + RNN-T lattice (never materialises (B,T,U+1,V) nor (B,T,U+1,2*O)).
 * Replaces JointNet.joint (networks/transducer.py:54-69) followed by RNNTLoss (model.py:39,57):
 *   z[b,t,u,:] = fc(gelu_tanh(cat(enc[b,t], dec[b,u]))) = A[b,t,:] + C[b,u,:] + bias     
 * Inputs here are the two small pre-GEMM results A = gelu(enc).W_e^T, C = gelu(dec).W_d^T (computed with
 * rnnt_hip_gemm_f32 + RNNT_GEMM_GELU_A) and fc.bias (V).  A(b,t,v) = A[b*a_sb + t*a_st + v],
 * C(b,u,v) = C[b*c_sb + u*c_su + v]  (so batch-major and time-major buffers both work, no copy).
 * Outputs: nll (B) = -log P(y|x) per utterance; dA, dC (same strides as A, C) = d(sum_b gscale*nll_b)/dA,dC.
 *   labels (B,U) int32 (U = U1-1), t_lens (B) int32 in [1,T], u_lens (B) int32 in [0,U].
 * ---------------------------------------------------------------------------------------------- */
size_t rnnt_hip_joint_loss_workspace_bytes(int32_t B, int32_t T, int32_t U1, int32_t V);
int rnnt_hip_joint_loss_fwd_bwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb, int64_t c_su,
                                const float* bias, const int32_t* labels, const int32_t* t_lens,
                                const int32_t* u_lens, int32_t B, int32_t T, int32_t U1, int32_t V, int32_t blank,
                                float gscale, float* nll, float* dA, float* dC, void* workspace,
                                size_t workspace_bytes, void* stream);

/* The same in two calls, for autograd: the forward is rnnt_hip_joint_loss_fwd_bwd with dA = dC = NULL (it leaves
 * log-softmax terms, alpha, beta and log Z in `workspace`); this runs the gradient kernels from that workspace on the SAME
 * A / C / bias / labels / lengths, with the upstream gradient per utterance: d(sum_b gscale * gvec[b * gvec_stride] * nll_b)/dA,dC
 * (gvec device, gvec_stride 1 = one value per utterance, 0 = ONE scalar for all; NULL = ones).  reduction="mean" of model.py:39
 * arrives here as gscale = 1/B with gvec = the 0-d gradient of the mean (stride 0). */
int rnnt_hip_joint_loss_bwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb, int64_t c_su,
                            const float* bias, const int32_t* labels, const int32_t* t_lens, const int32_t* u_lens,
                            int32_t B, int32_t T, int32_t U1, int32_t V, int32_t blank, float gscale, const float* gvec,
                            int32_t gvec_stride, float* dA, float* dC, void* workspace, size_t workspace_bytes, void* stream);
/* out[0] = scale * sum_i x[i] in a fixed order (reduction="mean" / "sum" of the per-utterance losses, model.py:39). */
int rnnt_hip_scaled_sum_f32(const float* x, int32_t n, float scale, float* out, void* stream);

/* Materialising joint for RNNTransducer.forward() (model.py:47-50): logits (B,T,U1,V) = A + C + bias. */
int rnnt_hip_joint_logits_fwd(const float* A, int64_t a_sb, int64_t a_st, const float* C, int64_t c_sb, int64_t c_su,
                              const float* bias, int32_t B, int32_t T, int32_t U1, int32_t V, float* logits,
                              void* stream);

/* warp-transducer-shaped entry (model.py:39,57): loss + gradient from dense logits (B,T,U1,V).
 * grad may be NULL (forward only).  grad = d(sum_b gscale*nll_b)/d logits. */
int rnnt_hip_loss_from_logits_fwd_bwd(const float* logits, const int32_t* labels, const int32_t* t_lens,
                                      const int32_t* u_lens, int32_t B, int32_t T, int32_t U1, int32_t V,
                                      int32_t blank, float gscale, float* nll, float* grad, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* Embedding forward (networks/decoder.py:69,102): out[m,:] = W[idx[m],:]  (row padding_idx of W is zero by
 * construction, nn.Embedding(padding_idx=blank)). */
int rnnt_hip_embedding_fwd(const float* W, const int64_t* idx, int64_t M, int32_t H, int32_t V, float* out, void* stream);

/* Same, for logits/grad stored as fp16 or bf16 (torchaudio's RNNTLoss takes half logits: model.py:28-31); the
 * log-softmax, alpha/beta and gradient arithmetic stay fp32/fp64, only loads/stores convert. */
#define RNNT_DTYPE_F32 0
#define RNNT_DTYPE_F16 1
#define RNNT_DTYPE_BF16 2
int rnnt_hip_loss_from_logits_fwd_bwd_ex(const void* logits, int32_t dtype, const int32_t* labels, const int32_t* t_lens,
                                         const int32_t* u_lens, int32_t B, int32_t T, int32_t U1, int32_t V, int32_t blank,
                                         float gscale, float* nll, void* grad, void* workspace, size_t workspace_bytes,
                                         void* stream);

/* One fused AdamW step over FLAT fp32 buffers (all parameters / gradients / moments of the module laid out back to back):
 * replaces torch.optim.AdamW's multi-tensor kernels at model.py:111-115.  Same update as torch (decoupled weight decay,
 * bias corrections from `step` >= 1). */
int rnnt_hip_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int64_t step, void* stream);
/* Same with g scaled by `grad_scale` on load (1/world of the data-parallel average: train.py:45 DDP semantics, folded into
 * the update instead of a separate pass over the gradients) and an optional device guard word: when *guard != 0 (the sticky
 * LSTM status word, see rnnt_lstm_desc.status) the kernel leaves p, m, v untouched. */
int rnnt_hip_adamw_step_ex(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, float grad_scale, const uint32_t* guard, void* stream);

/* column sums: out[n] = sum_m X[m*ld + n]  (bias gradients: fc.bias, out_proj.bias, LSTM biases).
 * Two-stage fixed-order reduction; workspace = rnnt_hip_colsum_workspace_bytes(M, N) bytes. */
size_t rnnt_hip_colsum_workspace_bytes(int64_t M, int64_t N);
int rnnt_hip_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);
/* out[n] += column sum (flat-gradient accumulation) */
int rnnt_hip_colsum_f32_acc(const float* X, int64_t M, int64_t N, int64_t ld, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Embedding backward (networks/decoder.py:69,102): dW[idx[m]] += dE[m] for idx[m] != padding_idx. dW (V,H)
 * must be zeroed by the caller. */
int rnnt_hip_embedding_bwd(const float* dE, const int64_t* idx, int64_t M, int32_t H, int32_t V, int64_t padding_idx,
                           float* dW, void* stream);
/* dW[v] += sum (row padding_idx untouched): accumulation into an existing gradient */
int rnnt_hip_embedding_bwd_acc(const float* dE, const int64_t* idx, int64_t M, int32_t H, int32_t V, int64_t padding_idx,
                               float* dW, void* stream);

/* Greedy decoding on device (replaces JointNet.recognize_greedy, networks/transducer.py:95-145, including its
 * single-step prediction-net call networks/decoder.py:121-123 and the 1-D joint networks/transducer.py:64-69).
 * One workgroup per utterance.  A = gelu(encoder_outputs) . fc.weight[:, :O_enc]^T + fc.bias for every frame
 * (time-major (T,B,V); computed by the caller with rnnt_hip_gemm_f32(RNNT_GEMM_GELU_A)).  Utterance b visits frames
 * t in [0, t_lens[b]) — the reference decodes one utterance per call, so its loop bound encoder_outputs.size(1) is that
 * utterance's own length; t_lens == NULL visits all T padded frames (what a batched reference call does).  Per frame: up to max_iters symbols; a symbol equal to the
 * last appended one still advances the prediction net but is not appended (transducer.py:132-137).
 * tokens (B,max_out) int64 (entries past ntok[b] are left untouched), ntok (B) int32; max_out >= T*max_iters never
 * truncates. */
#define RNNT_DECODE_MAX_LAYERS 8
typedef struct rnnt_decode_desc {
  int32_t T, B, V;       /* frames, utterances, vocabulary */
  int32_t Hp, O, L;      /* prediction-net hidden size (= embedding width), joint input width per side, layers */
  int32_t cell;          /* RNNT_CELL_* */
  int32_t blank, max_iters, max_out;
  const float* A;        /* (T,B,V) */
  const int32_t* t_lens; /* (B) device, or NULL */
  const float* emb;      /* (V,Hp)  decoder.embedding.weight */
  const float* w_ih[RNNT_DECODE_MAX_LAYERS]; /* (G*Hp,Hp) decoder.rnn.weight_ih_l{k} */
  const float* w_hh[RNNT_DECODE_MAX_LAYERS];
  const float* b_ih[RNNT_DECODE_MAX_LAYERS];
  const float* b_hh[RNNT_DECODE_MAX_LAYERS];
  const float* w_o;      /* (O,Hp)  decoder.out_proj.weight */
  const float* b_o;      /* (O) */
  const float* w_d;      /* fc.weight[:, O_enc:]  (V,O), row stride ld_d floats */
  int64_t ld_d;
  int64_t* tokens;
  int32_t* ntok;
} rnnt_decode_desc;
int rnnt_hip_greedy_decode(const rnnt_decode_desc* d, void* stream);

/* One prediction-net step for a batch with carried state (networks/decoder.py:121-123: `self.rnn(embedded,
 * prev_hidden_state)` on a (B,1) token column, as the reference's search loops call it).  h_in / c_in (L,B,Hp) may be NULL
 * (= zeros: prev_hidden_state None); h_out / c_out (L,B,Hp); the layer output is h_out[L-1].  c_* only for LSTM. */
typedef struct rnnt_prednet_step_desc {
  int32_t B, Hp, L, cell;
  const int64_t* tokens; /* (B) */
  const float* emb;      /* (V,Hp) */
  const float* w_ih[RNNT_DECODE_MAX_LAYERS];
  const float* w_hh[RNNT_DECODE_MAX_LAYERS];
  const float* b_ih[RNNT_DECODE_MAX_LAYERS];
  const float* b_hh[RNNT_DECODE_MAX_LAYERS];
  const float* h_in;
  const float* c_in;
  float* h_out;
  float* c_out;
} rnnt_prednet_step_desc;
int rnnt_hip_prednet_step(const rnnt_prednet_step_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input side on device (datamodule.py:48-90, done offline on the host by the reference).
 * rnnt_hip_frontend_norm_pad: per utterance b (row b of wav, lens[b] samples): optional mean / population-variance
 *   normalisation (datamodule.py:87-90), reflect padding by `pad` samples at the utterance's own ends (torch.stft
 *   center=True), zeros up to Lp.  out (B, Lp).
 * The windowed DFT is then ONE rnnt_hip_gemm_f32 over the frames in place: M = B*F rows with a_div = F, a_so = Lp,
 *   a_si = hop, K = n_fft, B = hann * [cos | -sin] basis (2*n_bins, n_fft).
 * rnnt_hip_power_mel_log1p: spec (M, 2*n_bins) = [re | im] -> out (M, n_mels) = log1p(fb^T |X|^2), rows whose frame index
 *   (m % frames_per_utt) is >= nframes[m / frames_per_utt] are written as 0 (the collate's pad value, dataloader.py:40).
 * ---------------------------------------------------------------------------------------------- */
int rnnt_hip_frontend_norm_pad(const float* wav, int64_t ld, const int32_t* lens, int32_t B, int32_t pad, int64_t Lp,
                               int32_t normalize, float* out, void* stream);
int rnnt_hip_power_mel_log1p(const float* spec, int64_t M, int32_t n_bins, const float* fb, int32_t n_mels,
                             const int32_t* nframes, int32_t frames_per_utt, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RNNT_HIP_H_ */
