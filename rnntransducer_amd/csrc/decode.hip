// On-device greedy RNN-T decoding.
//
// Replaces JointNet.recognize_greedy (networks/transducer.py:95-145): a host loop over frames with one `.item()` device
// sync per emitted symbol, a single-step prediction-net call (networks/decoder.py:121-123) and the 1-D joint
// (networks/transducer.py:64-69) per symbol.
//
// Semantics kept exactly: per utterance, for t in 0..t_lens[b]-1 (the reference decodes one utterance per call, so its
// encoder_outputs.size(1) IS that utterance's length; t_lens = null visits all T padded frames, which is what a batched
// reference call would do): up to `max_iters` times { tok = argmax_v joint(enc_t, dec); if tok == blank: stop this
// frame; append tok unless it equals the last appended token; advance the prediction net with tok }.
//
// Mapping: logits = A[t] + C with A = gelu(enc) W_e^T + bias for all frames (one hot-path GEMM, computed by the
// caller) and C = gelu(dec) W_d^T, which only changes when a symbol is emitted.  ONE workgroup (1024 threads) per
// utterance keeps the prediction-net state in LDS and streams the weights (L2 / Infinity-Cache resident, shared by
// all utterances) for each emitted symbol; the per-frame work is a V-wide argmax.  No inter-workgroup communication.
#include "common.hpp"

namespace rnnt {
namespace {

constexpr int DEC_THREADS = 1024;
constexpr int DEC_MAX_LAYERS = RNNT_DECODE_MAX_LAYERS;

struct DecodeK {
  int T, B, V, Hp, O, L, cell, blank, max_iters, max_out;
  const float* A;  // (T,B,V) time-major, bias included
  const int* t_lens;  // frames to visit per utterance, or null (= T)
  const float* emb;  // (V, Hp)
  const float* w_ih[DEC_MAX_LAYERS];
  const float* w_hh[DEC_MAX_LAYERS];
  const float* b_ih[DEC_MAX_LAYERS];
  const float* b_hh[DEC_MAX_LAYERS];
  const float* w_o;  // (O, Hp)
  const float* b_o;  // (O)
  const float* w_d;  // fc.weight[:, O_enc:] : (V, O) with row stride ld_d
  long ld_d;
  long long* tokens;  // (B, max_out)
  int* ntok;          // (B)
};

// y[r] = dot(W[r, :cols], x) (+ bias[r]) for r in [0, rows): one wave per group of RU rows (lanes along the contiguous k),
// 16 waves per pass.  All RU rows' loads are issued before any is consumed: a single row per wave keeps only 2 KB in
// flight per wave and the step becomes latency-bound (measured 308 us per prediction-net step at H=512).
constexpr int RU = 8;
__device__ __forceinline__ void matvec(const float* __restrict__ W, long ld, int rows, int cols, const float* __restrict__ x,
                                       float* __restrict__ y, const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = DEC_THREADS / 64;
  for (int r0 = wave * RU; r0 < rows; r0 += nw * RU) {
    float s[RU];
#pragma unroll
    for (int i = 0; i < RU; ++i) s[i] = 0.f;
    for (int k = 4 * lane; k < cols; k += 256) {
      f32x4 w[RU];
#pragma unroll
      for (int i = 0; i < RU; ++i) {
        const int r = r0 + i < rows ? r0 + i : rows - 1;  // clamp: tail rows re-read the last row, result discarded
        w[i] = *reinterpret_cast<const f32x4*>(W + (long)r * ld + k);
      }
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
#pragma unroll
      for (int i = 0; i < RU; ++i) s[i] += w[i][0] * xv[0] + w[i][1] * xv[1] + w[i][2] * xv[2] + w[i][3] * xv[3];
    }
#pragma unroll
    for (int i = 0; i < RU; ++i) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o);
    }
    if (lane < RU && r0 + lane < rows) {
      float v = s[0];
#pragma unroll
      for (int i = 1; i < RU; ++i) v = lane == i ? s[i] : v;
      y[r0 + lane] = v + (bias ? bias[r0 + lane] : 0.f);
    }
  }
}

// dynamic LDS: h[L][Hp] | c[L][Hp] | gi[4Hp] | gh[4Hp] | x[Hp] | dec[O] | Cv[V] | red (2 * 16 floats/ints) | ctl[4]
__global__ void __launch_bounds__(DEC_THREADS) greedy_decode_kernel(const DecodeK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Hp = p.Hp, O = p.O, V = p.V, L = p.L;
  float* h = reinterpret_cast<float*>(smem);
  float* c = h + L * Hp;
  float* gi = c + L * Hp;
  float* gh = gi + 4 * Hp;
  float* x = gh + 4 * Hp;
  float* dec = x + Hp;
  float* Cv = dec + O;
  float* redv = Cv + V;
  int* redi = reinterpret_cast<int*>(redv + 16);
  int* ctl = redi + 16;  // ctl[0] = token chosen this evaluation

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int NG = p.cell == RNNT_CELL_LSTM ? 4 : (p.cell == RNNT_CELL_GRU ? 3 : 1);

  for (int i = tid; i < 2 * L * Hp; i += DEC_THREADS) h[i] = 0.f;  // h and c (hidden_state = None -> zeros)
  __syncthreads();

  // one prediction-net step with input token `tok`, then C = gelu(out_proj(h_last)) . W_d^T
  auto prednet_step = [&](int tok) {
    for (int i = tid; i < Hp; i += DEC_THREADS) x[i] = p.emb[(long)tok * Hp + i];
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      matvec(p.w_ih[l], Hp, NG * Hp, Hp, x, gi, p.b_ih[l]);
      matvec(p.w_hh[l], Hp, NG * Hp, Hp, h + l * Hp, gh, p.b_hh[l]);
      __syncthreads();
      for (int j = tid; j < Hp; j += DEC_THREADS) {
        float hv;
        if (p.cell == RNNT_CELL_LSTM) {
          const float ig = sigmoidf_(gi[j] + gh[j]), fg = sigmoidf_(gi[Hp + j] + gh[Hp + j]);
          const float gg = tanhf(gi[2 * Hp + j] + gh[2 * Hp + j]), og = sigmoidf_(gi[3 * Hp + j] + gh[3 * Hp + j]);
          const float cv = fg * c[l * Hp + j] + ig * gg;
          c[l * Hp + j] = cv;
          hv = og * tanhf(cv);
        } else if (p.cell == RNNT_CELL_GRU) {
          const float rg = sigmoidf_(gi[j] + gh[j]), zg = sigmoidf_(gi[Hp + j] + gh[Hp + j]);
          const float ng = tanhf(gi[2 * Hp + j] + rg * gh[2 * Hp + j]);
          hv = (1.f - zg) * ng + zg * h[l * Hp + j];
        } else {
          const float pre = gi[j] + gh[j];
          hv = p.cell == RNNT_CELL_RNN_RELU ? fmaxf(pre, 0.f) : tanhf(pre);
        }
        h[l * Hp + j] = hv;
        x[j] = hv;  // input of the next layer (no dropout at inference)
      }
      __syncthreads();
    }
    matvec(p.w_o, Hp, O, Hp, h + (L - 1) * Hp, dec, p.b_o);
    __syncthreads();
    for (int i = tid; i < O; i += DEC_THREADS) dec[i] = gelu_tanh(dec[i]);
    __syncthreads();
    matvec(p.w_d, p.ld_d, V, O, dec, Cv, nullptr);
    __syncthreads();
  };

  // tok = argmax_v (A[t,b,v] + Cv[v]); lowest index among equal maxima (torch.argmax on a 1-D CPU/GPU tensor)
  auto frame_argmax = [&](int t) -> int {
    const float* a = p.A + ((long)t * p.B + b) * V;
    float best = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int v = tid; v < V; v += DEC_THREADS) {
      const float z = a[v] + Cv[v];
      if (z > best || (z == best && v < bi)) { best = z; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { redv[wave] = best; redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < DEC_THREADS / 64; ++w)
        if (redv[w] > best || (redv[w] == best && redi[w] < bi)) { best = redv[w]; bi = redi[w]; }
      ctl[0] = bi;
    }
    __syncthreads();
    const int tok = ctl[0];
    __syncthreads();
    return tok;
  };

  prednet_step(p.blank);  // decoder_input = [[blank]] (transducer.py:118-119)
  int n = 0;
  long long last = p.blank;  // pred_tokens = [blank]
  int Tb = p.t_lens ? p.t_lens[b] : p.T;
  Tb = Tb < 0 ? 0 : (Tb > p.T ? p.T : Tb);
  for (int t = 0; t < Tb; ++t) {
    for (int u = 0; u < p.max_iters; ++u) {
      const int tok = frame_argmax(t);
      if (tok == p.blank) break;
      if (last != tok) {
        if (n < p.max_out && tid == 0) p.tokens[(long)b * p.max_out + n] = tok;
        ++n;
        last = tok;
      }
      prednet_step(tok);
    }
  }
  if (tid == 0) p.ntok[b] = n < p.max_out ? n : p.max_out;
}


// One prediction-net step for a batch (networks/decoder.py:121-123: `self.rnn(embedded, prev_hidden_state)` on a (B,1) token
// column): one workgroup per batch row, state through LDS, same matvec / cell code as the search kernel.
struct StepK {
  int B, Hp, L, cell;
  const long long* tokens;  // (B)
  const float* emb;
  const float* w_ih[DEC_MAX_LAYERS];
  const float* w_hh[DEC_MAX_LAYERS];
  const float* b_ih[DEC_MAX_LAYERS];
  const float* b_hh[DEC_MAX_LAYERS];
  const float* h_in;  // (L,B,Hp) or null (zeros)
  const float* c_in;  // LSTM only; or null
  float* h_out;       // (L,B,Hp)
  float* c_out;       // LSTM only
};

__global__ void __launch_bounds__(DEC_THREADS) prednet_step_kernel(const StepK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Hp = p.Hp, L = p.L;
  float* h = reinterpret_cast<float*>(smem);
  float* c = h + L * Hp;
  float* gi = c + L * Hp;
  float* gh = gi + 4 * Hp;
  float* x = gh + 4 * Hp;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int NG = p.cell == RNNT_CELL_LSTM ? 4 : (p.cell == RNNT_CELL_GRU ? 3 : 1);
  for (int i = tid; i < L * Hp; i += DEC_THREADS) {
    const int l = i / Hp, j = i % Hp;
    h[i] = p.h_in ? p.h_in[((long)l * p.B + b) * Hp + j] : 0.f;
    c[i] = (p.c_in && p.cell == RNNT_CELL_LSTM) ? p.c_in[((long)l * p.B + b) * Hp + j] : 0.f;
  }
  const long long tok = p.tokens[b];
  for (int i = tid; i < Hp; i += DEC_THREADS) x[i] = p.emb[tok * Hp + i];
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    matvec(p.w_ih[l], Hp, NG * Hp, Hp, x, gi, p.b_ih[l]);
    matvec(p.w_hh[l], Hp, NG * Hp, Hp, h + l * Hp, gh, p.b_hh[l]);
    __syncthreads();
    for (int j = tid; j < Hp; j += DEC_THREADS) {
      float hv;
      if (p.cell == RNNT_CELL_LSTM) {
        const float ig = sigmoidf_(gi[j] + gh[j]), fg = sigmoidf_(gi[Hp + j] + gh[Hp + j]);
        const float gg = tanhf(gi[2 * Hp + j] + gh[2 * Hp + j]), og = sigmoidf_(gi[3 * Hp + j] + gh[3 * Hp + j]);
        const float cv = fg * c[l * Hp + j] + ig * gg;
        c[l * Hp + j] = cv;
        hv = og * tanhf(cv);
      } else if (p.cell == RNNT_CELL_GRU) {
        const float rg = sigmoidf_(gi[j] + gh[j]), zg = sigmoidf_(gi[Hp + j] + gh[Hp + j]);
        const float ng = tanhf(gi[2 * Hp + j] + rg * gh[2 * Hp + j]);
        hv = (1.f - zg) * ng + zg * h[l * Hp + j];
      } else {
        const float pre = gi[j] + gh[j];
        hv = p.cell == RNNT_CELL_RNN_RELU ? fmaxf(pre, 0.f) : tanhf(pre);
      }
      h[l * Hp + j] = hv;
      x[j] = hv;
    }
    __syncthreads();
  }
  for (int i = tid; i < L * Hp; i += DEC_THREADS) {
    const int l = i / Hp, j = i % Hp;
    p.h_out[((long)l * p.B + b) * Hp + j] = h[i];
    if (p.c_out && p.cell == RNNT_CELL_LSTM) p.c_out[((long)l * p.B + b) * Hp + j] = c[i];
  }
}

}  // namespace
}  // namespace rnnt

using namespace rnnt;

extern "C" int rnnt_hip_greedy_decode(const rnnt_decode_desc* d, void* stream) {
  RNNT_CHECK_ARG(d != nullptr, "greedy_decode: null descriptor");
  RNNT_CHECK_ARG(d->T >= 1 && d->B >= 1 && d->V >= 1 && d->Hp >= 4 && d->Hp % 4 == 0 && d->O >= 4 && d->O % 4 == 0,
                 "greedy_decode: bad dims (hidden and output sizes must be multiples of 4)");
  RNNT_CHECK_ARG(d->L >= 1 && d->L <= DEC_MAX_LAYERS, "greedy_decode: 1..%d prediction-net layers", DEC_MAX_LAYERS);
  RNNT_CHECK_ARG(d->cell >= RNNT_CELL_LSTM && d->cell <= RNNT_CELL_RNN_RELU, "greedy_decode: unknown cell type");
  RNNT_CHECK_ARG(d->blank >= 0 && d->blank < d->V && d->max_iters >= 1 && d->max_out >= 1, "greedy_decode: bad blank/max_iters/max_out");
  RNNT_CHECK_ARG(d->A && d->emb && d->w_o && d->b_o && d->w_d && d->tokens && d->ntok, "greedy_decode: null pointer");
  RNNT_CHECK_ARG(d->ld_d % 4 == 0 && (reinterpret_cast<uintptr_t>(d->w_d) & 15) == 0, "greedy_decode: fc slice must be 16-byte aligned");
  DecodeK k;
  k.T = d->T; k.B = d->B; k.V = d->V; k.Hp = d->Hp; k.O = d->O; k.L = d->L; k.cell = d->cell; k.blank = d->blank;
  k.max_iters = d->max_iters; k.max_out = d->max_out;
  k.A = d->A; k.t_lens = d->t_lens; k.emb = d->emb;
  for (int l = 0; l < d->L; ++l) {
    RNNT_CHECK_ARG(d->w_ih[l] && d->w_hh[l] && d->b_ih[l] && d->b_hh[l], "greedy_decode: null weight (layer %d)", l);
    k.w_ih[l] = d->w_ih[l]; k.w_hh[l] = d->w_hh[l]; k.b_ih[l] = d->b_ih[l]; k.b_hh[l] = d->b_hh[l];
  }
  k.w_o = d->w_o; k.b_o = d->b_o; k.w_d = d->w_d; k.ld_d = d->ld_d;
  k.tokens = (long long*)d->tokens; k.ntok = d->ntok;
  const size_t lds = ((size_t)2 * d->L * d->Hp + 8 * d->Hp + d->Hp + d->O + d->V + 32 + 8) * 4;
  RNNT_CHECK_ARG(lds <= 160 * 1024, "greedy_decode: state needs %zu B of LDS (> 160 KiB)", lds);
  if (lds > 64 * 1024)
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)greedy_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope prof(RNNT_K_MISC, 4.0 * (double)d->T * d->B * d->V, (hipStream_t)stream);
  hipLaunchKernelGGL(greedy_decode_kernel, dim3(d->B), dim3(DEC_THREADS), lds, (hipStream_t)stream, k);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

extern "C" int rnnt_hip_prednet_step(const rnnt_prednet_step_desc* d, void* stream) {
  RNNT_CHECK_ARG(d != nullptr, "prednet_step: null descriptor");
  RNNT_CHECK_ARG(d->B >= 1 && d->Hp >= 4 && d->Hp % 4 == 0, "prednet_step: bad dims (hidden size must be a multiple of 4)");
  RNNT_CHECK_ARG(d->L >= 1 && d->L <= DEC_MAX_LAYERS, "prednet_step: 1..%d layers", DEC_MAX_LAYERS);
  RNNT_CHECK_ARG(d->cell >= RNNT_CELL_LSTM && d->cell <= RNNT_CELL_RNN_RELU, "prednet_step: unknown cell type");
  RNNT_CHECK_ARG(d->tokens && d->emb && d->h_out && (d->c_out || d->cell != RNNT_CELL_LSTM), "prednet_step: null pointer");
  StepK k;
  k.B = d->B; k.Hp = d->Hp; k.L = d->L; k.cell = d->cell;
  k.tokens = (const long long*)d->tokens; k.emb = d->emb;
  for (int l = 0; l < d->L; ++l) {
    RNNT_CHECK_ARG(d->w_ih[l] && d->w_hh[l] && d->b_ih[l] && d->b_hh[l], "prednet_step: null weight (layer %d)", l);
    k.w_ih[l] = d->w_ih[l]; k.w_hh[l] = d->w_hh[l]; k.b_ih[l] = d->b_ih[l]; k.b_hh[l] = d->b_hh[l];
  }
  k.h_in = d->h_in; k.c_in = d->c_in; k.h_out = d->h_out; k.c_out = d->c_out;
  const size_t lds = ((size_t)2 * d->L * d->Hp + 9 * d->Hp) * 4;
  RNNT_CHECK_ARG(lds <= 160 * 1024, "prednet_step: state needs %zu B of LDS (> 160 KiB)", lds);
  if (lds > 64 * 1024)
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)prednet_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope prof(RNNT_K_MISC, 8.0 * (double)d->L * d->B * d->Hp, (hipStream_t)stream);
  hipLaunchKernelGGL(prednet_step_kernel, dim3(d->B), dim3(DEC_THREADS), lds, (hipStream_t)stream, k);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}
