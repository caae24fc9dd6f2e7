/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (rnntransducer_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the RNN-T loss the reference calls at model.py:57 through
 * `Warp_RNNTLoss(blank, reduction="mean")` (constructed model.py:39) /
 * `Torch_RNNTLoss(blank, reduction="mean")` (model.py:31).  The arithmetic itself lives in third-party
 * packages that are NOT under /root/reference (warprnnt_pytorch from the warp-transducer fork named in
 * README.md:9-10, branch espnet_v1.1, no commit pin; torchaudio, unpinned) and neither is installed here,
 * so this follows the published algorithm (Graves 2012, arXiv:1211.3711, cited README.md:3) with the
 * warp-transducer conventions written out in SURVEY.md Appendix A.3:
 *   - row-wise log-softmax over V of the raw joint logits,
 *   - alpha/beta log-space recursions over the T_b x (U_b+1) lattice,
 *   - NLL_b = -(alpha[T_b-1,U_b] + lp_blank[T_b-1,U_b]) = -beta[0,0],
 *   - gradient w.r.t. the *logits* (softmax fused), zero outside the valid lattice,
 *   - per-utterance values returned; the caller applies reduction="mean" (sum/B).
 *
 * PARITY PIN: the reference holds no test or golden vector for this boundary.
 * This file is pinned by (tests/test_oracle_loss.py): the upstream 2x3x5 known-answer vector G3
 * (NLL 4.495666 and its 30 gradient entries), the upstream B=2 x T=4 x U=2 x V=3 vector (costs 4.28065286 / 3.93843698), brute-force enumeration of all monotone lattice paths,
 * and torch-autograd through an independent float64 DP.  It is NOT pinned by running the reference's
 * own loss package: that part of parity is "unpinned by the reference".
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC (see oracle/build_oracle.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

#define IDX3(t, u, U1) ((size_t)(t) * (size_t)(U1) + (size_t)(u))

#define DEFINE_RNNT_REF(NAME, REAL, EXP, LOG1P, FABS, NEGINF)                                           \
  static inline REAL NAME##_logaddexp(REAL a, REAL b) {                                                \
    if (a == NEGINF) return b;                                                                         \
    if (b == NEGINF) return a;                                                                         \
    REAL m = a > b ? a : b;                                                                            \
    return m + LOG1P(EXP(-FABS(a - b)));                                                               \
  }                                                                                                    \
  /* logits: (B,T,U1,V) contiguous; labels: (B,U1-1); returns 0, or -1 on bad lengths, -2 on OOM */    \
  int NAME(const REAL* logits, const int* labels, const int* t_lens, const int* u_lens, int B, int T,  \
           int U1, int V, int blank, REAL* nll, REAL* grad) {                                          \
    int status = 0;                                                                                    \
    for (int b = 0; b < B; ++b)                                                                        \
      if (t_lens[b] < 1 || t_lens[b] > T || u_lens[b] < 0 || u_lens[b] + 1 > U1) return -1;            \
    _Pragma("omp parallel for schedule(dynamic)") for (int b = 0; b < B; ++b) {                        \
      const int Tb = t_lens[b], Ub1 = u_lens[b] + 1;                                                   \
      const REAL* z = logits + (size_t)b * T * U1 * V;                                                 \
      const int* y = labels + (size_t)b * (U1 - 1);                                                    \
      REAL* lse = (REAL*)malloc(sizeof(REAL) * (size_t)T * U1 * 3);                                    \
      REAL* al = (REAL*)malloc(sizeof(REAL) * (size_t)T * U1 * 2);                                     \
      if (!lse || !al) {                                                                               \
        status = -2;                                                                                   \
        free(lse);                                                                                     \
        free(al);                                                                                      \
        continue;                                                                                      \
      }                                                                                                \
      REAL* blk = lse + (size_t)T * U1;                                                                \
      REAL* emt = blk + (size_t)T * U1;                                                                \
      REAL* be = al + (size_t)T * U1;                                                                  \
      /* log-softmax pieces per valid cell: lse, log p(blank), log p(y_u) */                           \
      for (int t = 0; t < Tb; ++t)                                                                     \
        for (int u = 0; u < Ub1; ++u) {                                                                \
          const REAL* zz = z + IDX3(t, u, U1) * V;                                                     \
          REAL m = zz[0];                                                                              \
          for (int v = 1; v < V; ++v) m = zz[v] > m ? zz[v] : m;                                       \
          REAL s = 0;                                                                                  \
          for (int v = 0; v < V; ++v) s += EXP(zz[v] - m);                                             \
          REAL l = m + (REAL)log((double)s);                                                           \
          lse[IDX3(t, u, U1)] = l;                                                                     \
          blk[IDX3(t, u, U1)] = zz[blank] - l;                                                         \
          emt[IDX3(t, u, U1)] = (u < Ub1 - 1) ? zz[y[u]] - l : NEGINF;                                 \
        }                                                                                              \
      /* alpha */                                                                                      \
      for (int t = 0; t < Tb; ++t)                                                                     \
        for (int u = 0; u < Ub1; ++u) {                                                                \
          REAL a;                                                                                      \
          if (t == 0 && u == 0) a = 0;                                                                 \
          else {                                                                                       \
            REAL no_emit = (t > 0) ? al[IDX3(t - 1, u, U1)] + blk[IDX3(t - 1, u, U1)] : NEGINF;        \
            REAL emit = (u > 0) ? al[IDX3(t, u - 1, U1)] + emt[IDX3(t, u - 1, U1)] : NEGINF;           \
            a = NAME##_logaddexp(no_emit, emit);                                                       \
          }                                                                                            \
          al[IDX3(t, u, U1)] = a;                                                                      \
        }                                                                                              \
      /* beta */                                                                                       \
      for (int t = Tb - 1; t >= 0; --t)                                                                \
        for (int u = Ub1 - 1; u >= 0; --u) {                                                           \
          REAL v;                                                                                      \
          if (t == Tb - 1 && u == Ub1 - 1) v = blk[IDX3(t, u, U1)];                                    \
          else {                                                                                       \
            REAL no_emit = (t < Tb - 1) ? be[IDX3(t + 1, u, U1)] + blk[IDX3(t, u, U1)] : NEGINF;       \
            REAL emit = (u < Ub1 - 1) ? be[IDX3(t, u + 1, U1)] + emt[IDX3(t, u, U1)] : NEGINF;         \
            v = NAME##_logaddexp(no_emit, emit);                                                       \
          }                                                                                            \
          be[IDX3(t, u, U1)] = v;                                                                      \
        }                                                                                              \
      const REAL logZ = be[0];                                                                         \
      nll[b] = -logZ;                                                                                  \
      if (grad) {                                                                                      \
        REAL* g = grad + (size_t)b * T * U1 * V;                                                       \
        for (size_t i = 0; i < (size_t)T * U1 * V; ++i) g[i] = 0;                                      \
        for (int t = 0; t < Tb; ++t)                                                                   \
          for (int u = 0; u < Ub1; ++u) {                                                              \
            const REAL* zz = z + IDX3(t, u, U1) * V;                                                   \
            REAL* gg = g + IDX3(t, u, U1) * V;                                                         \
            const REAL a = al[IDX3(t, u, U1)], l = lse[IDX3(t, u, U1)];                                \
            const REAL occ = EXP(a + be[IDX3(t, u, U1)] - logZ);                                       \
            for (int v = 0; v < V; ++v) gg[v] = occ * EXP(zz[v] - l);                                  \
            if (t == Tb - 1 && u == Ub1 - 1) gg[blank] -= EXP(a + blk[IDX3(t, u, U1)] - logZ);         \
            if (t < Tb - 1) gg[blank] -= EXP(a + blk[IDX3(t, u, U1)] + be[IDX3(t + 1, u, U1)] - logZ); \
            if (u < Ub1 - 1)                                                                           \
              gg[y[u]] -= EXP(a + emt[IDX3(t, u, U1)] + be[IDX3(t, u + 1, U1)] - logZ);                \
          }                                                                                            \
      }                                                                                                \
      free(lse);                                                                                       \
      free(al);                                                                                        \
    }                                                                                                  \
    return status;                                                                                     \
  }

DEFINE_RNNT_REF(rnnt_loss_ref_f64, double, exp, log1p, fabs, (-INFINITY))
DEFINE_RNNT_REF(rnnt_loss_ref_f32, float, expf, log1pf, fabsf, (-INFINITY))

/*
 * CPU build of the C-ABI entry `rnnt_hip_loss_from_logits_fwd_bwd` (include/rnnt_hip.h; it replaces the reference's call
 * `Warp_RNNTLoss.__call__` at model.py:57): SAME name, SAME argument list, host pointers instead of device pointers; `workspace`,
 * `workspace_bytes` and `stream` are accepted and ignored.  Tests load this library and librnnt_hip.so side by side and hand both the
 * same arguments ("a CPU build of the same ABI (the restatement) is exported from a second .so for tests").
 * Arithmetic: the float64 restatement above, outputs rounded to float once; grad = d(sum_b gscale * nll_b) / d logits.
 * Returns 0, or -1 (RNNT_ERR_INVALID) on bad arguments / lengths, as the HIP entry does.
 */
int rnnt_hip_loss_from_logits_fwd_bwd(const float* logits, const int* labels, const int* t_lens, const int* u_lens, int B, int T,
                                      int U1, int V, int blank, float gscale, float* nll, float* grad, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  if (!logits || !t_lens || !u_lens || !nll || B < 1 || T < 1 || U1 < 1 || V < 1 || blank < 0 || blank >= V || (U1 > 1 && !labels))
    return -1;
  const size_t n = (size_t)B * T * U1 * V;
  double* z = (double*)malloc(sizeof(double) * n);
  double* g = grad ? (double*)malloc(sizeof(double) * n) : NULL;
  double* l = (double*)malloc(sizeof(double) * (size_t)B);
  if (!z || !l || (grad && !g)) {
    free(z); free(g); free(l);
    return -2;
  }
  for (size_t i = 0; i < n; ++i) z[i] = (double)logits[i];
  int rc = rnnt_loss_ref_f64(z, labels, t_lens, u_lens, B, T, U1, V, blank, l, g);
  if (rc == 0) {
    for (int b = 0; b < B; ++b) nll[b] = (float)l[b];
    if (grad)
      for (size_t i = 0; i < n; ++i) grad[i] = (float)(g[i] * (double)gscale);
  }
  free(z); free(g); free(l);
  return rc == 0 ? 0 : -1;
}
