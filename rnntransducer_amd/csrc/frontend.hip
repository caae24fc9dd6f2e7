// On-device input side: what the reference does offline on the host per utterance in
// datamodule.py -- mean_var_norm (:87-90), torchaudio MelSpectrogram(n_fft = win = 400, hop = 160, n_mels = 80, center,
// reflect pad, power 2, HTK mel scale, no norm) (:48-66), log1p (:67) and the (mel,time)->(time,mel) transpose.
//
//   1. frontend_norm_pad_kernel : per utterance mean / population variance (two passes), normalise, reflect-pad by n_fft/2
//                                 at the utterance's OWN ends, zero the rest of the row
//   2. rnnt_hip_gemm_f32        : windowed DFT as a GEMM, frames addressed in place through the operand row map
//                                 (row (b,f) starts at b*Lp + f*hop: no unfold copy); basis = hann * [cos | -sin] (402 x 400)
//   3. power_mel_log1p_kernel   : |X|^2 -> mel filterbank -> log1p, frames beyond an utterance's count written as 0
//                                 (the collate's padding value, dataloader.py:40)
#include "common.hpp"

namespace rnnt {
namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// one workgroup per utterance
__global__ void __launch_bounds__(1024) frontend_norm_pad_kernel(const float* __restrict__ wav, long ld, const int* __restrict__ lens,
                                                                  int P, long Lp, int normalize, float* __restrict__ out) {
  __shared__ double red[16];
  const int b = blockIdx.x, L = lens[b];
  const float* x = wav + (long)b * ld;
  float* y = out + (long)b * Lp;
  float mean = 0.f, rstd = 1.f;
  if (normalize && L > 0) {
    double s = 0.0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) s += x[i];
    const double m = block_sum(s, red) / L;
    double q = 0.0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      const double dlt = x[i] - m;
      q += dlt * dlt;
    }
    const double var = block_sum(q, red) / L;  // np.var: population variance (datamodule.py:89)
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-7));
  }
  for (long i = threadIdx.x; i < Lp; i += blockDim.x) {
    long j = i - P;
    float v = 0.f;
    if (L > 0 && i < (long)L + 2 * P) {
      if (j < 0) j = -j;                       // reflect (no edge repeat), torch.stft(center=True, pad_mode="reflect")
      if (j >= L) j = 2l * (L - 1) - j;
      if (j >= 0 && j < L) v = (x[j] - mean) * rstd;
    }
    y[i] = v;
  }
}

constexpr int FR = 16;  // frames per workgroup

// spec (M, 2*NB): [re_0..re_{NB-1} | im_0..im_{NB-1}] per frame; fb (NB, NM); out (M, NM)
__global__ void __launch_bounds__(256) power_mel_log1p_kernel(const float* __restrict__ spec, long M, int NB, const float* __restrict__ fb,
                                                               int NM, const int* __restrict__ nframes, int F, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* fbs = reinterpret_cast<float*>(smem);   // [NB][NM]
  float* pw = fbs + NB * NM;                      // [FR][NB]
  const long m0 = (long)blockIdx.x * FR;
  for (int i = threadIdx.x; i < NB * NM; i += 256) fbs[i] = fb[i];
  for (int i = threadIdx.x; i < FR * NB; i += 256) {
    const int f = i / NB, j = i % NB;
    float p = 0.f;
    if (m0 + f < M) {
      const float re = spec[(m0 + f) * 2 * NB + j], im = spec[(m0 + f) * 2 * NB + NB + j];
      p = re * re + im * im;
    }
    pw[i] = p;
  }
  __syncthreads();
  const int f = threadIdx.x >> 4, n0 = threadIdx.x & 15;
  const long m = m0 + f;
  if (m >= M) return;
  const bool live = (int)(m % F) < nframes[m / F];
  for (int n = n0; n < NM; n += 16) {
    float acc = 0.f;
    if (live) {
      const float* prow = pw + f * NB;
      for (int j = 0; j < NB; ++j) acc += prow[j] * fbs[j * NM + n];
    }
    out[m * NM + n] = live ? log1pf(acc) : 0.f;
  }
}

}  // namespace
}  // namespace rnnt

using namespace rnnt;

extern "C" int rnnt_hip_frontend_norm_pad(const float* wav, int64_t ld, const int32_t* lens, int32_t B, int32_t pad, int64_t Lp,
                                          int32_t normalize, float* out, void* stream) {
  RNNT_CHECK_ARG(wav && lens && out, "frontend_norm_pad: null pointer");
  RNNT_CHECK_ARG(B >= 1 && pad >= 0 && Lp >= 2 * (int64_t)pad + 1 && ld >= 1, "frontend_norm_pad: bad dims");
  ProfScope prof(RNNT_K_MISC, 4.0 * ((double)B * ld + (double)B * Lp), (hipStream_t)stream);
  hipLaunchKernelGGL(frontend_norm_pad_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, wav, (long)ld, lens, pad, (long)Lp,
                     normalize, out);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}

extern "C" int rnnt_hip_power_mel_log1p(const float* spec, int64_t M, int32_t n_bins, const float* fb, int32_t n_mels,
                                        const int32_t* nframes, int32_t frames_per_utt, float* out, void* stream) {
  RNNT_CHECK_ARG(spec && fb && nframes && out, "power_mel_log1p: null pointer");
  RNNT_CHECK_ARG(M >= 1 && n_bins >= 1 && n_mels >= 1 && frames_per_utt >= 1 && M % frames_per_utt == 0, "power_mel_log1p: bad dims");
  const size_t lds = ((size_t)n_bins * n_mels + (size_t)FR * n_bins) * 4;
  RNNT_CHECK_ARG(lds <= 160 * 1024, "power_mel_log1p: filterbank does not fit LDS (%zu B)", lds);
  if (lds > 64 * 1024)
    RNNT_CHECK_HIP(hipFuncSetAttribute((const void*)power_mel_log1p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope prof(RNNT_K_MISC, 4.0 * ((double)M * 2 * n_bins + (double)M * n_mels), (hipStream_t)stream);
  hipLaunchKernelGGL(power_mel_log1p_kernel, dim3((unsigned)ceil_div(M, FR)), dim3(256), lds, (hipStream_t)stream, spec, (long)M, n_bins,
                     fb, n_mels, nframes, frames_per_utt, out);
  RNNT_CHECK_LAUNCH();
  return RNNT_OK;
}
