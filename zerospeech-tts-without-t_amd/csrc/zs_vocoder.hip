// zs_vocoder.hip -- batched Griffin-Lim pieces (reference convert.py:39-62 on librosa stft/istft):
// 1024-point FFTs held entirely in LDS, one 256-thread workgroup per (utterance, frame).
//   zs_gl_istft        : spec frame -> Hermitian extension -> inverse FFT -> * hann(800 padded to 1024)
//                        -> frames_ws;  then overlap-add / window-sum-square / centre trim -> wav
//   zs_gl_stft_project : wav -> reflect-padded frame * window -> FFT -> E ; spec = mag * E / max(1e-8, |E|)
// The FFT is a radix-2 Stockham autosort (10 stages, ping-pong between two LDS buffers, natural-order
// output, twiddle table of 512 entries built once per workgroup with sincospif).  The spectrogram of an
// utterance (4.1 KB/frame) stays in L2/Infinity Cache across the 300 iterations: FFT/latency-bound.
#include "zs_common.h"

namespace {

constexpr int NFFT = 1024, HOP = 200, WIN = 800, NBIN = 513, NTV = 256;

__device__ __forceinline__ float hann_padded(int n) {
  // scipy get_window('hann', 800, fftbins=True) centre-padded to 1024 (librosa.util.pad_center)
  const int lpad = (NFFT - WIN) / 2;
  const int j = n - lpad;
  if (j < 0 || j >= WIN) return 0.f;
  return 0.5f - 0.5f * cospif(2.0f * (float)j / (float)WIN);
}

// in-place-by-ping-pong FFT of 1024 complex points in LDS.  a holds the input; returns pointer to result.
// sign = -1 forward (exp(-i..)), +1 inverse (unscaled).
__device__ __forceinline__ float2* fft1024(float2* a, float2* b, const float2* tw, int tid, float sign) {
  float2* x = a;
  float2* y = b;
  int s = 1;
#pragma unroll 1
  for (int n = NFFT; n > 1; n >>= 1, s <<= 1) {
    const int m = n >> 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = tid + NTV * r;          // butterfly index in [0, 512)
      const int p = i / s, q = i - p * s;   // s is a power of two; m*s == 512
      float2 w = tw[p * s];
      w.y *= sign * -1.0f;                  // table holds exp(-2 pi i j/1024): (cos, -sin); inverse flips the sign
      const float2 u = x[q + s * p];
      const float2 v = x[q + s * (p + m)];
      const float2 d = make_float2(u.x - v.x, u.y - v.y);
      y[q + s * (2 * p)] = make_float2(u.x + v.x, u.y + v.y);
      y[q + s * (2 * p + 1)] = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
  }
  return x;
}

__device__ __forceinline__ void build_twiddles(float2* tw, int tid) {
  for (int j = tid; j < NFFT / 2; j += NTV) {
    float sn, cs;
    sincospif(2.0f * (float)j / (float)NFFT, &sn, &cs);
    tw[j] = make_float2(cs, -sn);
  }
}

__global__ __launch_bounds__(NTV) void gl_iframes_kernel(const ZsGlIstft p) {
  __shared__ float2 bufA[NFFT];
  __shared__ float2 bufB[NFFT];
  __shared__ float2 tw[NFFT / 2];
  const int tid = threadIdx.x, t = blockIdx.x, u = blockIdx.y;
  if (t >= p.lengths[u]) return;                                   // uniform per workgroup
  build_twiddles(tw, tid);
  const float2* S = reinterpret_cast<const float2*>(p.spec) + ((int64_t)u * p.T_max + t) * NBIN;
  for (int k = tid; k < NFFT; k += NTV) {                           // Hermitian extension (irfft semantics)
    float2 v;
    if (k <= NFFT / 2) { v = S[k]; if (k == 0 || k == NFFT / 2) v.y = 0.f; }
    else { v = S[NFFT - k]; v.y = -v.y; }
    bufA[k] = v;
  }
  __syncthreads();
  const float2* r = fft1024(bufA, bufB, tw, tid, +1.0f);
  float* out = p.frames_ws + ((int64_t)u * p.T_max + t) * NFFT;
  for (int n = tid; n < NFFT; n += NTV) out[n] = r[n].x * (1.0f / NFFT) * hann_padded(n);
}

__global__ void gl_ola_kernel(const ZsGlIstft p) {
  const int u = blockIdx.y;
  const int T = p.lengths[u];
  const int L = HOP * (T - 1);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= L) return;
  const int pos = n + NFFT / 2;
  int i0 = (pos - (NFFT - 1) + HOP - 1) / HOP; if (i0 < 0) i0 = 0;
  int i1 = pos / HOP; if (i1 > T - 1) i1 = T - 1;
  const float* fr = p.frames_ws + (int64_t)u * p.T_max * NFFT;
  float acc = 0.f, wss = 0.f;
  for (int i = i0; i <= i1; ++i) {                                  // ascending frame order, as the reference loop
    const int k = pos - HOP * i;
    acc += fr[(int64_t)i * NFFT + k];
    const float w = hann_padded(k);
    wss += w * w;
  }
  if (wss > 1.17549435e-38f) acc /= wss;                           // util.tiny(float32)
  p.wav[(int64_t)u * p.wav_ld + n] = acc;
}

__global__ __launch_bounds__(NTV) void gl_stft_project_kernel(const ZsGlStft p) {
  __shared__ float2 bufA[NFFT];
  __shared__ float2 bufB[NFFT];
  __shared__ float2 tw[NFFT / 2];
  const int tid = threadIdx.x, t = blockIdx.x, u = blockIdx.y;
  const int T = p.lengths[u];
  if (t >= T) return;
  build_twiddles(tw, tid);
  const int L = HOP * (T - 1);
  const float* w = p.wav + (int64_t)u * p.wav_ld;
  for (int n = tid; n < NFFT; n += NTV) {
    int idx = t * HOP + n - NFFT / 2;                               // np.pad(y, 512, mode='reflect')
    if (idx < 0) idx = -idx;
    if (idx >= L) idx = 2 * (L - 1) - idx;
    bufA[n] = make_float2(w[idx] * hann_padded(n), 0.f);
  }
  __syncthreads();
  const float2* E = fft1024(bufA, bufB, tw, tid, -1.0f);
  const float* M = p.mag + ((int64_t)u * p.T_max + t) * NBIN;
  float2* S = reinterpret_cast<float2*>(p.spec) + ((int64_t)u * p.T_max + t) * NBIN;
  for (int k = tid; k < NBIN; k += NTV) {
    const float2 e = E[k];
    const float a = sqrtf(e.x * e.x + e.y * e.y);
    const float sc = M[k] / fmaxf(1e-8f, a);                        // X_best = spectrogram * est / max(1e-8, |est|)
    S[k] = make_float2(e.x * sc, e.y * sc);
  }
}

// preprocess.py:227-258 (get_spectrograms) after the host-side trim: pre-emphasis y[n] - c*y[n-1], librosa.stft(n_fft 1024,
// hop 200, hann(800) centred in the 1024 frame, center=True reflect padding), |.|, 20*log10(max(1e-5, .)), normalise and
// clip to [1e-8, 1].  One workgroup per (utterance, frame); amp (optional) keeps the linear magnitudes for the mel product.
__global__ __launch_bounds__(NTV) void pre_spectrogram_kernel(const ZsPreSpec p) {
  __shared__ float2 bufA[NFFT];
  __shared__ float2 bufB[NFFT];
  __shared__ float2 tw[NFFT / 2];
  const int tid = threadIdx.x, t = blockIdx.x, u = blockIdx.y;
  const int L = p.n_samples[u];
  const int T = 1 + L / HOP;                                        // librosa: 1 + len(y) // hop_length frames
  if (t >= T || L < 2) return;
  build_twiddles(tw, tid);
  const float* w = p.wav + (int64_t)u * p.wav_ld;
  for (int n = tid; n < NFFT; n += NTV) {
    int idx = t * HOP + n - NFFT / 2;                               // np.pad(y, 512, mode='reflect') of the pre-emphasised signal
    if (idx < 0) idx = -idx;
    if (idx >= L) idx = 2 * (L - 1) - idx;
    idx = min(max(idx, 0), L - 1);
    const float y = idx > 0 ? w[idx] - p.preemph * w[idx - 1] : w[0];
    bufA[n] = make_float2(y * hann_padded(n), 0.f);
  }
  __syncthreads();
  const float2* E = fft1024(bufA, bufB, tw, tid, -1.0f);
  float* out = p.mag + ((int64_t)u * p.T_max + t) * p.mag_ld;
  float* amp = p.amp ? p.amp + ((int64_t)u * p.T_max + t) * NBIN : nullptr;
  for (int k = tid; k < NBIN; k += NTV) {
    const float2 e = E[k];
    const float a = sqrtf(e.x * e.x + e.y * e.y);
    if (amp) amp[k] = a;
    const float db = 20.f * log10f(fmaxf(1e-5f, a));
    out[k] = fminf(fmaxf((db - p.ref_db + p.max_db) / p.max_db, 1e-8f), 1.f);
  }
}

// mel[t][m] = clip((20 log10(max(1e-5, sum_k basis[m][k] amp[t][k])) - ref + max) / max, 1e-8, 1)   (preprocess.py:243-252)
__global__ void pre_mel_kernel(const float* amp, const float* basis, float* mel, int64_t rows, int n_mels, float ref_db, float max_db) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  for (int m = threadIdx.x; m < n_mels; m += blockDim.x) {
    const float* a = amp + r * NBIN;
    const float* b = basis + (int64_t)m * NBIN;
    float s = 0.f;
    for (int k = 0; k < NBIN; ++k) s += b[k] * a[k];
    const float db = 20.f * log10f(fmaxf(1e-5f, s));
    mel[r * n_mels + m] = fminf(fmaxf((db - ref_db + max_db) / max_db, 1e-8f), 1.f);
  }
}

__global__ void gl_denorm_kernel(const float* in, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = fminf(fmaxf(in[i], 0.f), 1.f) * 100.f - 100.f + 20.f;   // convert.py:57
    out[i] = powf(10.0f, m * 0.05f);                                          // convert.py:58
  }
}

}  // namespace

extern "C" int zs_gl_istft(const ZsGlIstft* p, void* stream) {
  ZS_REQUIRE(p && p->spec && p->lengths && p->wav && p->frames_ws && p->n_utt > 0 && p->T_max > 1, "zs_gl_istft: bad args");
  ZS_REQUIRE(p->wav_ld >= (int64_t)HOP * (p->T_max - 1), "zs_gl_istft: wav_ld too small");
  hipLaunchKernelGGL(gl_iframes_kernel, dim3(p->T_max, p->n_utt), dim3(NTV), 0, (hipStream_t)stream, *p);
  int rc = zs_check_launch("zs_gl_istft.frames");
  if (rc) return rc;
  const int L = HOP * (p->T_max - 1);
  hipLaunchKernelGGL(gl_ola_kernel, dim3((L + 255) / 256, p->n_utt), dim3(256), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_gl_istft.ola");
}

extern "C" int zs_gl_stft_project(const ZsGlStft* p, void* stream) {
  ZS_REQUIRE(p && p->wav && p->mag && p->lengths && p->spec && p->n_utt > 0 && p->T_max > 1, "zs_gl_stft_project: bad args");
  hipLaunchKernelGGL(gl_stft_project_kernel, dim3(p->T_max, p->n_utt), dim3(NTV), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_gl_stft_project");
}

extern "C" int zs_pre_spectrogram(const ZsPreSpec* p, void* stream) {
  ZS_REQUIRE(p && p->wav && p->n_samples && p->mag && p->n_utt > 0 && p->T_max > 0 && p->mag_ld >= NBIN && p->max_db > 0.f,
             "zs_pre_spectrogram: bad args");
  hipLaunchKernelGGL(pre_spectrogram_kernel, dim3(p->T_max, p->n_utt), dim3(NTV), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_pre_spectrogram");
}

extern "C" int zs_pre_mel(const float* amp, const float* basis, float* mel, int64_t rows, int32_t n_mels, float ref_db, float max_db,
                          void* stream) {
  ZS_REQUIRE(amp && basis && mel && rows > 0 && n_mels > 0 && max_db > 0.f, "zs_pre_mel: bad args");
  ZS_REQUIRE(rows < (1ll << 31), "zs_pre_mel: too many rows");
  hipLaunchKernelGGL(pre_mel_kernel, dim3((unsigned)rows), dim3(128), 0, (hipStream_t)stream, amp, basis, mel, rows, (int)n_mels, ref_db, max_db);
  return zs_check_launch("zs_pre_mel");
}

extern "C" int zs_gl_denormalize(const float* mag_norm, float* mag_amp, int64_t n, void* stream) {
  ZS_REQUIRE(mag_norm && mag_amp && n > 0, "zs_gl_denormalize: bad args");
  int64_t nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(gl_denorm_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, mag_norm, mag_amp, n);
  return zs_check_launch("zs_gl_denormalize");
}
