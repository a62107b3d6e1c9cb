// zs_griffin.hip -- one Griffin-Lim iteration as ONE kernel (reference convert.py:39-52 on librosa stft / istft):
//     x = istft(X);  E = stft(x);  X' = S * E / max(1e-8, |E|)
// fused per utterance tile so that neither the windowed frames nor the waveform ever leave LDS.
//
// gfx950 design
//  * A workgroup (4 waves) owns a tile of F consecutive STFT frames [t0, t1) of one utterance.  With the hann(800) window
//    centred in the 1024-sample frame a waveform sample is touched by exactly 4 frames and an output frame depends on the
//    input frames t-3 .. t+3: the tile inverse-transforms F + 6 frames (3 halo frames per side), overlap-adds them into an LDS segment of (F-1)*200 + 800 samples, divides by the window-sum-square, and
//    transforms its own F frames forward.  HBM / Infinity-Cache traffic per frame and iteration: the spectrogram once in
//    and once out (2 x 4.1 KB) + the halo re-reads -- against ~25 KB for the unfused frames -> overlap-add -> frames chain.
//  * Frames whose indices are equal mod 4 never overlap (4 * hop = 800 = window length), so the overlap-add runs in four
//    phases, one residue each, with plain non-atomic LDS accumulation: deterministic, no race.
//  * The transform is a 512-point complex FFT per WAVE (real 1024-point transform through the even/odd packing: half the
//    butterflies of the complex 1024-point FFT): three radix-8 Stockham passes with the 8 points of a butterfly in registers,
//    two wave-private LDS exchanges between them (skewed by i + i/8 -> every ds access pattern of the three passes is
//    bank-conflict free for a half-wave), no workgroup barrier inside the transform.  Twiddles: one LDS table of the
//    1024th roots of unity per workgroup (the 512th and 64th roots are strided views), powers w^2..w^7 by multiplication.
//  * Twiddle table, window, exchange buffers and segment take 37 KB of LDS at F = 10 with 4 waves -> four workgroups (16 waves)
//    per CU; F = 26 with 8 waves: 68 KB -> two workgroups, the same 16 waves, 1.23 instead of 1.6 inverse transforms per frame.
//    By the counters (profiles/r03_gl_counters.txt) the round-2 kernel spent 67 % of the vector-issue slots (it is NOT bound by
//    the memory system: 2.7 TB/s): round 3 removed 26 % of its vector instructions (interior 1/window-sum-square table, wave-
//    uniform fast paths without per-sample bounds checks, rsqrt projection) and requests a wave's next spectrum right behind
//    the current transform: 36.9 -> 29.2 ms for 300 iterations of 64 utterances of 200..700 frames.
// zs_griffin_lim runs the whole loop (n_iter launches ping-ponging two spectrogram buffers + the final inverse pass) from
// one C call.  The older one-transform-per-workgroup kernels (zs_vocoder.hip) remain as the variant the tests compare with.
#include <atomic>
#include <chrono>
#include <mutex>

#include "zs_common.h"

namespace {

constexpr int NB = 513, HOP = 200, WLEN = 800, WOFF = 112, HALF = 512;
constexpr int WBUF = 576;      // complex slots of a wave's exchange buffer: 512 + 512/8 skew
constexpr int WTAB = 520;      // 513 roots, padded
constexpr int GL_TILE_LARGE = 26;
constexpr int GL_TILE_DEFAULT = 10;   // measured on 64 utterances of 200..700 frames: 10 -> 43 ms, 26 -> 51 ms, 42 -> 67 ms per 300 iterations

__device__ __forceinline__ int padi(int i) { return i + (i >> 3); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// LDS hand-off between the lanes of ONE wave: LDS instructions of a wave execute in order, so a later ds_read sees an earlier
// ds_write of any lane; the wait + barrier only stop the compiler from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// 8-point DFT in registers, natural-order output.  INV = false: kernel exp(-2 pi i rq/8); true: exp(+2 pi i rq/8).
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  constexpr float S = 0.70710678118654752f;
  const float2 t0 = cadd(v[0], v[4]), t1 = csub(v[0], v[4]);
  const float2 t2 = cadd(v[2], v[6]), t3 = csub(v[2], v[6]);
  const float2 t4 = cadd(v[1], v[5]), t5 = csub(v[1], v[5]);
  const float2 t6 = cadd(v[3], v[7]), t7 = csub(v[3], v[7]);
  const float2 e0 = cadd(t0, t2), e1 = csub(t0, t2), e2 = cadd(t4, t6), e3 = csub(t4, t6);
  // w4 * z : forward -i z = (z.y, -z.x); inverse +i z = (-z.y, z.x)
  const float2 w4e3 = INV ? make_float2(-e3.y, e3.x) : make_float2(e3.y, -e3.x);
  v[0] = cadd(e0, e2); v[4] = csub(e0, e2);
  v[2] = cadd(e1, w4e3); v[6] = csub(e1, w4e3);
  const float2 u1 = INV ? make_float2(S * (t5.x - t5.y), S * (t5.x + t5.y)) : make_float2(S * (t5.x + t5.y), S * (t5.y - t5.x));
  const float2 u2 = INV ? make_float2(-t3.y, t3.x) : make_float2(t3.y, -t3.x);
  const float2 u3 = INV ? make_float2(S * (-t7.x - t7.y), S * (t7.x - t7.y)) : make_float2(S * (t7.y - t7.x), S * (-t7.x - t7.y));
  const float2 o0 = cadd(t1, u2), o1 = csub(t1, u2), o2 = cadd(u1, u3), o3 = csub(u1, u3);
  const float2 w4o3 = INV ? make_float2(-o3.y, o3.x) : make_float2(o3.y, -o3.x);
  v[1] = cadd(o0, o2); v[5] = csub(o0, o2);
  v[3] = cadd(o1, w4o3); v[7] = csub(o1, w4o3);
}

// v[r] *= w1^r, r = 1..7
__device__ __forceinline__ void twiddle8(float2 (&v)[8], float2 w1) {
  const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
  v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
  v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
}

// 512-point complex FFT of one wave.  Lane j holds elements j + 64 r (r = 0..7) on entry and the natural-order result in the
// same layout on exit.  Stockham radix-8: pass with Ns = 1, 8, 64: butterfly j reads x[j + 64 r], multiplies by
// exp(-+2 pi i r (j mod Ns) / (8 Ns)), and writes y[(j / Ns) 8 Ns + (j mod Ns) + r Ns].  W = exp(-2 pi i k / 1024).
template <bool INV>
__device__ __forceinline__ void fft512(float2 (&v)[8], float2* buf, const float2* W, int lane) {
  dft8<INV>(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[9 * lane + r] = v[r];                    // padi(8 lane + r)
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[padi(lane + 64 * r)];
  wave_lds_sync();
  {
    float2 w1 = W[16 * (lane & 7)];                                        // 64th roots
    if (INV) w1.y = -w1.y;
    twiddle8(v, w1);
  }
  dft8<INV>(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[padi(((lane >> 3) << 6) + (lane & 7) + 8 * r)] = v[r];
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[padi(lane + 64 * r)];
  wave_lds_sync();
  {
    float2 w1 = W[2 * lane];                                               // 512th roots
    if (INV) w1.y = -w1.y;
    twiddle8(v, w1);
  }
  dft8<INV>(v);
}

// exp(-2 pi i k / 1024), k = 0..512, and hann(800) (periodic): filled by gl_tables_kernel at the head of every zs_griffin_lim /
// zs_gl_iter call (stream-ordered before the kernels that read it; concurrent calls write identical values)
__device__ float2 g_gl_roots[WTAB];
__device__ float g_gl_window[WLEN];

__global__ void gl_tables_kernel() {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= HALF) {
    float sn, cs;
    sincospif((float)j * (1.0f / 512.0f), &sn, &cs);
    g_gl_roots[j] = make_float2(cs, -sn);
  }
  if (j < WLEN) g_gl_window[j] = 0.5f - 0.5f * cospif(2.0f * (float)j / (float)WLEN);
}

__device__ __forceinline__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// PF: request the spectrum of a wave's next inverse frame right behind the current transform (see below).  NWAVE: waves per
// workgroup (4 or 8): a tile of F frames inverse-transforms F + 6 frames in four phases, so the phases are balanced over the waves
// when F + 6 is a multiple of 4 NWAVE -- F = 10 with 4 waves (1.6 inverse transforms per frame), F = 26 with 8 waves (1.23).
template <bool PF, int NWAVE, bool FROM_MAG>
__global__ __launch_bounds__(64 * NWAVE) void gl_iter_kernel(const ZsGlIter p, const float* spec_in, float* spec_out, int F) {
  constexpr int NT_ = 64 * NWAVE;
  extern __shared__ __align__(16) unsigned char gl_smem[];
  float2* W = reinterpret_cast<float2*>(gl_smem);
  float* win = reinterpret_cast<float*>(W + WTAB);
  float* iwss = win + WLEN;                                                   // [HOP]: 1 / window-sum-square of interior samples
  float2* wb = reinterpret_cast<float2*>(iwss + HOP);
  float* seg = reinterpret_cast<float*>(wb + NWAVE * WBUF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = blockIdx.y;
  const int T = p.lengths[u];
  if (T < 4) return;                                                         // reflect padding of 512 needs >= 513 samples
  const int nt = (T + F - 1) / F, tile = blockIdx.x;
  if (tile >= nt) return;
  const int t0 = (int)((int64_t)tile * T / nt), t1 = (int)((int64_t)(tile + 1) * T / nt);   // balanced tiles, each >= 2 frames
  const int L = HOP * (T - 1);
  const int n_lo = max(0, t0 * HOP - 400), n_hi = min(L, (t1 - 1) * HOP + 400);
  const int nseg = n_hi - n_lo;
  for (int j = tid; j <= HALF; j += NT_) W[j] = g_gl_roots[j];
  for (int j = tid; j < WLEN; j += NT_) win[j] = g_gl_window[j];                                          // hann(800), periodic
  for (int j = tid; j < nseg; j += NT_) seg[j] = 0.f;
  __syncthreads();
  // A sample covered by four whole frames (everything but the first / last 400 samples of an utterance) sees the window-sum-square
  // sum_m win^2(rho + 200 m), rho = (pos - WOFF) mod HOP, summed in the same order as the general loop below: 200 values per
  // workgroup instead of a 4-term loop and a division per sample (read after the barriers of the overlap-add phases)
  for (int j = tid; j < HOP; j += NT_) {
    const int tid_ = j;
    float wss = 0.f;
#pragma unroll
    for (int m = 3; m >= 0; --m) { const float w = win[tid_ + HOP * m]; wss += w * w; }
    iwss[tid_] = 1.0f / wss;
  }

  // ---- inverse transforms of frames i_lo..i_hi, overlap-added into seg (four phases: i mod 4) --------------------------
  const int i_lo = max(0, floordiv(n_lo - 400, HOP) + 1), i_hi = min(T - 1, (n_hi - 1 + 400) / HOP);
  float2* mybuf = wb + wave * WBUF;
  const float2* Sin = reinterpret_cast<const float2*>(spec_in) + (int64_t)u * p.T_max * NB;
  // Frames of this wave: phase ph takes the frames i = i_lo + ((ph - i_lo) & 3) + 4 wave + 16 j.  PF: the spectrum of the wave's
  // NEXT frame (in this or a later phase) is requested right after the current transform, so its latency runs under the
  // overlap-add, the phase barrier and the other waves' work -- and not during the transform, where the registers are needed.
  float2 xa[8], xb[8];
  const float* Min = p.mag + (int64_t)u * p.T_max * NB;
  auto fetch = [&](int fi) {
    if constexpr (FROM_MAG) {                                                  // first iteration: X0 = S, zero phase (convert.py:41)
      const float* Mx = Min + (int64_t)fi * NB;
#pragma unroll
      for (int r = 0; r < 8; ++r) { xa[r] = make_float2(Mx[lane + 64 * r], 0.f); xb[r] = make_float2(Mx[HALF - lane - 64 * r], 0.f); }
      return;
    }
    const float2* X = Sin + (int64_t)fi * NB;
#pragma unroll
    for (int r = 0; r < 8; ++r) { xa[r] = X[lane + 64 * r]; xb[r] = X[HALF - lane - 64 * r]; }
  };
  auto first_of = [&](int ph) { return i_lo + ((ph - i_lo) & 3) + 4 * wave; };
  if (PF) {
    int ph0 = 0;
    while (ph0 < 4 && first_of(ph0) > i_hi) ++ph0;
    if (ph0 < 4) fetch(first_of(ph0));
  }
#pragma unroll 1
  for (int ph = 0; ph < 4; ++ph) {
#pragma unroll 1
    for (int i = first_of(ph); i <= i_hi; i += 4 * NWAVE) {
      float2 v[8];
      if (!PF) fetch(i);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        float2 a = xa[r], b = xb[r];
        if (k == 0) { a.y = 0.f; b.y = 0.f; }                                // irfft ignores the imaginary parts of DC / Nyquist
        b.y = -b.y;
        const float2 s = cadd(a, b), d = csub(a, b);
        const float2 w = W[k];
        const float2 t = cmul(d, make_float2(w.x, -w.y));                    // d * exp(+2 pi i k / 1024)
        v[r] = make_float2(s.x - t.y, s.y + t.x);                            // Z = s + i t
      }
      fft512<true>(v, mybuf, W, lane);
      if (PF) {                                                              // successor of (ph, i) in this wave's sequence
        int nph = ph, ni = i + 4 * NWAVE;
        while (ni > i_hi && ++nph < 4) ni = first_of(nph);
        if (nph < 4) fetch(ni);
      }
      const int nb = i * HOP - HALF;
      constexpr float SC = 1.0f / 1024.0f;
      if (nb + WOFF >= n_lo && nb + WOFF + WLEN <= n_hi) {                     // (wave-uniform) the whole windowed frame lies in the segment
        float* sp = seg + (nb - n_lo);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int k = 2 * (lane + 64 * r);
          if ((r > 0 || k >= WOFF) && (r < 7 || k < WOFF + WLEN)) {          // WOFF and WLEN are even: k + 1 is inside with k
            const float2 w2 = *reinterpret_cast<const float2*>(win + (k - WOFF));
            sp[k] += v[r].x * SC * w2.x;
            sp[k + 1] += v[r].y * SC * w2.y;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int k = 2 * (lane + 64 * r);
          const int n0 = nb + k;
          if (k >= WOFF && k < WOFF + WLEN && n0 >= n_lo && n0 < n_hi) seg[n0 - n_lo] += v[r].x * SC * win[k - WOFF];
          if (k + 1 >= WOFF && k + 1 < WOFF + WLEN && n0 + 1 >= n_lo && n0 + 1 < n_hi)
            seg[n0 + 1 - n_lo] += v[r].y * SC * win[k + 1 - WOFF];
        }
      }
    }
    __syncthreads();
  }
  // ---- window-sum-square normalisation (librosa.istft) --------------------------------------------------------------------
  for (int j = tid; j < nseg; j += NT_) {
    const int pos = n_lo + j + HALF;
    const int q = pos - WOFF;                                                  // >= 400
    const int ib_raw = q / HOP, ia_raw = ib_raw - 3;                          // frames WOFF <= pos - i HOP < WOFF + WLEN
    float val = seg[j];
    if (ia_raw >= 0 && ib_raw <= T - 1) {
      val *= iwss[q - ib_raw * HOP];
    } else {
      const int ia = max(0, ia_raw), ib = min(T - 1, ib_raw);
      float wss = 0.f;
      for (int i = ia; i <= ib; ++i) { const float w = win[pos - i * HOP - WOFF]; wss += w * w; }
      if (wss > 1.17549435e-38f) val /= wss;
    }
    seg[j] = val;
  }
  __syncthreads();

  if (spec_out == nullptr) {                                                  // final pass: the waveform is the result
    const int own_lo = tile == 0 ? 0 : t0 * HOP - 100, own_hi = tile == nt - 1 ? L : t1 * HOP - 100;
    float* wav = p.wav + (int64_t)u * p.wav_ld;
    for (int n = own_lo + tid; n < own_hi; n += NT_) wav[n] = seg[n - n_lo];
    return;
  }

  // ---- forward transforms of the tile's own frames + projection onto the given magnitudes -----------------------------------
  const float* Mu = p.mag + (int64_t)u * p.T_max * NB;
  float2* Sout = reinterpret_cast<float2*>(spec_out) + (int64_t)u * p.T_max * NB;
#pragma unroll 1
  for (int t = t0 + wave; t < t1; t += NWAVE) {
    float2 v[8];
    const int nb = t * HOP - HALF;
    if (nb + WOFF >= n_lo && nb + WOFF + WLEN <= n_hi) {                       // (wave-uniform) no reflection at the utterance's ends:
      const float* sp = seg + (nb - n_lo);                                    // every sample of the windowed frame is in the segment
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = 2 * (lane + 64 * r);
        v[r] = make_float2(0.f, 0.f);
        if ((r > 0 || k >= WOFF) && (r < 7 || k < WOFF + WLEN)) {
          const float2 w2 = *reinterpret_cast<const float2*>(win + (k - WOFF));
          v[r] = make_float2(sp[k] * w2.x, sp[k + 1] * w2.y);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = 2 * (lane + 64 * r);
        float x0 = 0.f, x1 = 0.f;
        if (k >= WOFF && k < WOFF + WLEN) {                                     // k even, WOFF even: k + 1 is inside too
          int a = nb + k, b = nb + k + 1;
          if (a < 0) a = -a;
          if (a >= L) a = 2 * (L - 1) - a;
          if (b < 0) b = -b;
          if (b >= L) b = 2 * (L - 1) - b;
          a = min(max(a, n_lo), n_hi - 1); b = min(max(b, n_lo), n_hi - 1);     // (no-ops: the segment covers the reflections)
          x0 = seg[a - n_lo] * win[k - WOFF];
          x1 = seg[b - n_lo] * win[k + 1 - WOFF];
        }
        v[r] = make_float2(x0, x1);
      }
    }
    const float* M = Mu + (int64_t)t * NB;
    float mg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) mg[r] = M[lane + 64 * r];                     // in flight under the transform
    const float mg_ny = M[HALF];
    fft512<false>(v, mybuf, W, lane);
#pragma unroll
    for (int r = 0; r < 8; ++r) mybuf[padi(lane + 64 * r)] = v[r];
    wave_lds_sync();
    float2* So = Sout + (int64_t)t * NB;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = lane + 64 * r;
      const float2 a = v[r];
      float2 b = mybuf[padi((HALF - k) & (HALF - 1))];
      b.y = -b.y;
      const float2 s = cadd(a, b), d = csub(a, b);
      const float2 tt = cmul(d, W[k]);
      const float2 e = make_float2(0.5f * (s.x + tt.y), 0.5f * (s.y - tt.x));   // E[k] = (a+b)/2 - (i/2) W^k (a-b)
      // X = S * E / max(1e-8, |E|)   (convert.py:50) as S * E * rsqrt(max(1e-16, |E|^2)): one v_rsq_f32 (1 ulp) instead of a
      // square root and a division (the projection was a fifth of the kernel's vector instructions)
      const float sc = mg[r] * __builtin_amdgcn_rsqf(fmaxf(1e-16f, e.x * e.x + e.y * e.y));
      So[k] = make_float2(e.x * sc, e.y * sc);
    }
    if (lane == 0) {                                                          // Nyquist bin: E[512] = Re Z0 - Im Z0
      const float2 z0 = mybuf[0];
      const float e = z0.x - z0.y;
      const float sc = mg_ny / fmaxf(1e-8f, fabsf(e));
      So[HALF] = make_float2(e * sc, 0.f);
    }
    wave_lds_sync();
  }
}

// signal.lfilter([1], [1, -coef], wav) (convert.py:60), float64 recursion y[n] = x[n] + coef y[n-1]: one workgroup per
// utterance, every thread a contiguous chunk; pass 1 finds each chunk's end value from a zero state, a serial scan over the
// 256 chunk ends gives every chunk its true carry-in, pass 2 replays the chunk from it (same operation order as the
// sequential filter inside a chunk; the carry-in differs from the sequential value at the 1e-16 level).
__global__ __launch_bounds__(256) void gl_deemph_scan_kernel(float* wav, int64_t wav_ld, const int32_t* lengths, float coef) {
  __shared__ double e_end[256];
  __shared__ double carry[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int L = HOP * (lengths[u] - 1);
  if (L <= 0) return;
  float* w = wav + (int64_t)u * wav_ld;
  const int C = (L + 255) / 256;
  const int lo = min(L, tid * C), hi = min(L, lo + C);
  const double a = (double)coef;
  double acc = 0.0;
  for (int i = lo; i < hi; ++i) acc = (double)w[i] + a * acc;
  e_end[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    double aC = 1.0;
    for (int i = 0; i < C; ++i) aC *= a;
    double c = 0.0;
    for (int j = 0; j < 256; ++j) {
      carry[j] = c;                                                          // y[j*C - 1]
      const int n = min(L, (j + 1) * C) - min(L, j * C);
      c = (n == C) ? e_end[j] + aC * c : (n > 0 ? e_end[j] + pow(a, (double)n) * c : c);
    }
  }
  __syncthreads();
  acc = carry[tid];
  for (int i = lo; i < hi; ++i) { acc = (double)w[i] + a * acc; w[i] = (float)acc; }
}

// mean square of the centred frames of librosa.effects.trim (zs_gl_frame_mse): one workgroup per (frame, utterance)
__global__ __launch_bounds__(256) void gl_frame_mse_kernel(const float* wav, int64_t wav_ld, const int32_t* lengths, int frame_length, int hop,
                                                          double* mse, int64_t mse_ld) {
  __shared__ double red[256];
  const int u = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
  const int L = HOP * (lengths[u] - 1);
  if (L <= frame_length / 2 || f > L / hop) return;
  const float* w = wav + (int64_t)u * wav_ld;
  double acc = 0.0;
  for (int i = tid; i < frame_length; i += 256) {
    int j = f * hop + i - frame_length / 2;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    const double v = (double)w[j];
    acc += v * v;
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) mse[(int64_t)u * mse_ld + f] = red[0] / (double)frame_length;
}

std::atomic<int> g_gl_prefetch{1};
std::once_flag g_gl_lds_once;

size_t gl_lds_bytes(int F, int nwave) { return (size_t)WTAB * 8 + WLEN * 4 + HOP * 4 + (size_t)nwave * WBUF * 8 + ((size_t)(F - 1) * HOP + 800) * 4; }

void gl_set_lds_attr();

int gl_launch(const ZsGlIter* p, const float* in, float* out, hipStream_t s, int from_mag = 0) {
  // default tile: 26 frames on 8 waves once the batch fills the chip (>= 8192 frames: 1.23 instead of 1.6 inverse transforms per
  // frame, 29.2 against 30.9 ms for 300 iterations of 64 utterances of 200..700 frames), else 10 frames on 4 waves (more workgroups)
  const int F = p->tile_frames > 0 ? p->tile_frames : ((int64_t)p->n_utt * p->T_max >= 8192 ? GL_TILE_LARGE : GL_TILE_DEFAULT);
  dim3 grid((unsigned)((p->T_max + F - 1) / F), (unsigned)p->n_utt);
  const bool pf = g_gl_prefetch.load(std::memory_order_relaxed) != 0;
#define ZS_GL_LAUNCH(PFV, NW, FM) hipLaunchKernelGGL((gl_iter_kernel<PFV, NW, FM>), grid, dim3(64 * NW), gl_lds_bytes(F, NW), s, *p, in, out, F)
  // FROM_MAG (the first iteration reads X0 = S from the magnitudes) is a template parameter: as a run-time branch it took the
  // steady-state kernel over 128 VGPRs, i.e. from two workgroups per CU to one (26 -> 39 ms)
  if (F > 16) {                                                               // large tiles: 8 waves share the segment
    std::call_once(g_gl_lds_once, gl_set_lds_attr);
    if (from_mag) { if (pf) ZS_GL_LAUNCH(true, 8, true); else ZS_GL_LAUNCH(false, 8, true); }
    else { if (pf) ZS_GL_LAUNCH(true, 8, false); else ZS_GL_LAUNCH(false, 8, false); }
  } else if (from_mag) { if (pf) ZS_GL_LAUNCH(true, 4, true); else ZS_GL_LAUNCH(false, 4, true); }
  else { if (pf) ZS_GL_LAUNCH(true, 4, false); else ZS_GL_LAUNCH(false, 4, false); }
#undef ZS_GL_LAUNCH
  return zs_check_launch("zs_gl_iter");
}

void gl_set_lds_attr() {                                                      // 8-wave tiles of F = 42 frames need 79 KB of LDS
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_kernel<true, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_kernel<false, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_kernel<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_kernel<false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

int gl_check(const ZsGlIter* p, const char* what) {
  ZS_REQUIRE(p && p->mag && p->lengths && p->n_utt > 0 && p->T_max >= 4, "%s: bad args", what);
  ZS_REQUIRE(p->tile_frames == 0 || (p->tile_frames >= 4 && p->tile_frames <= 42), "%s: tile_frames must be in [4, 42]", what);
  return ZS_OK;
}

}  // namespace

// "gl_prefetch" / "gl_chains" knobs of zs_set_option
int zs_gl_prefetch_option(int value) { return g_gl_prefetch.exchange(value ? 1 : 0, std::memory_order_relaxed); }

extern "C" int zs_gl_iter(const ZsGlIter* p, void* stream) {
  int rc = gl_check(p, "zs_gl_iter");
  if (rc) return rc;
  hipLaunchKernelGGL(gl_tables_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream);
  ZS_REQUIRE(p->spec_in && p->spec_in != p->spec_out, "zs_gl_iter: spec_in must be given and differ from spec_out");
  ZS_REQUIRE(p->spec_out || (p->wav && p->wav_ld >= (int64_t)HOP * (p->T_max - 1)), "zs_gl_iter: final pass needs wav (wav_ld >= 200*(T_max-1))");
  return gl_launch(p, p->spec_in, p->spec_out, (hipStream_t)stream);
}

namespace {

// The iterations of different utterances are independent, and a launch over ALL utterances ends in a partly filled last round of
// workgroups (64 utterances of 200..700 frames: 1130 tiles of 26 frames on 512 resident workgroups = 2.2 rounds; 22-frame tiles
// take 39.5 ms where 23-frame tiles take 32.0).  zs_griffin_lim therefore runs the loop as `gl_chains` independent launch chains
// (contiguous utterance ranges of equal utterance count) on side streams: a chain's partial rounds are filled
// by the other chains' workgroups -- no iteration-wide barrier across utterances that do not depend on each other.
constexpr int GL_MAX_CHAINS = 4, GL_CANDIDATES = 10;
struct GlSide {
  hipStream_t s[GL_MAX_CHAINS] = {};
  int n = 0;                                   // mutually concurrent streams found
  hipEvent_t fork = nullptr, join[GL_MAX_CHAINS] = {};
  bool ready = false;
};
std::mutex g_gl_side_mu;
GlSide g_gl_side[16];
std::atomic<int> g_gl_chains_used{1};
std::atomic<int> g_gl_chains{3};   // measured on 64 utterances of 200..700 frames: 1 chain 29.7, 2: 26.0, 3: 25.7, 4: 32.0 ms per 300 iterations

// one wave busy for `ticks` of the 100 MHz wall clock (bounded: the loop ends when the clock has advanced)
__global__ void gl_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// Do kernels launched on a and b run side by side?  ROCm multiplexes a process's streams onto a few hardware queues (four by
// default, least-used first), and launches of two streams that share one execute in order -- which streams share depends on
// everything the process created before (torch's pool, the trainer's side streams).  Measured, not guessed: two 200-us spin
// kernels take ~0.2 ms when the streams are independent and ~0.4 ms when they are not.
bool gl_concurrent(hipStream_t a, hipStream_t b) {
  constexpr long long TICKS = 20000;           // 200 us
  int ok = 0;
  for (int rep = 0; rep < 2; ++rep) {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(gl_spin_kernel, dim3(1), dim3(64), 0, a, TICKS);
    hipLaunchKernelGGL(gl_spin_kernel, dim3(1), dim3(64), 0, b, TICKS);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ok += us < 330.0;
  }
  return ok == 2;
}

GlSide* gl_side() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_gl_side_mu);
  GlSide& g = g_gl_side[dev];
  if (!g.ready) {
    g.ready = true;                            // one attempt per device and process; n stays 0 if anything fails
    if (hipEventCreateWithFlags(&g.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < GL_MAX_CHAINS; ++i)
      if (hipEventCreateWithFlags(&g.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    hipStream_t cand[GL_CANDIDATES] = {};
    for (int i = 0; i < GL_CANDIDATES; ++i)
      if (hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;      // once per process: the probe below needs a quiet device
    bool used[GL_CANDIDATES] = {};
    for (int i = 0; i < GL_CANDIDATES && g.n < GL_MAX_CHAINS; ++i) {
      bool indep = true;
      for (int k = 0; k < g.n && indep; ++k) indep = gl_concurrent(g.s[k], cand[i]);
      if (indep) { g.s[g.n++] = cand[i]; used[i] = true; }
    }
    for (int i = 0; i < GL_CANDIDATES; ++i)
      if (!used[i]) (void)hipStreamDestroy(cand[i]);
  }
  return g.n >= 2 ? &g : nullptr;
}

int gl_chain(const ZsGlIter& q, float* spec_a, float* spec_b, int n_iter, hipStream_t s) {
  float* cur = spec_a;
  float* nxt = spec_b;
  for (int it = 0; it < n_iter; ++it) {
    int rc = gl_launch(&q, cur, nxt, s, it == 0);
    if (rc) return rc;
    float* t = cur; cur = nxt; nxt = t;
  }
  return gl_launch(&q, cur, nullptr, s, n_iter == 0);
}

}  // namespace

int zs_gl_chains_option(int value) { return g_gl_chains.exchange(value < 1 ? 1 : (value > GL_MAX_CHAINS ? GL_MAX_CHAINS : value), std::memory_order_relaxed); }

extern "C" int zs_gl_chains_used(void) { return g_gl_chains_used.load(std::memory_order_relaxed); }

extern "C" int zs_griffin_lim(const ZsGlIter* p, float* spec_a, float* spec_b, int32_t n_iter, void* stream) {
  int rc = gl_check(p, "zs_griffin_lim");
  if (rc) return rc;
  ZS_REQUIRE(spec_a && spec_b && spec_a != spec_b && n_iter >= 0, "zs_griffin_lim: two distinct spectrogram buffers are required");
  ZS_REQUIRE(p->wav && p->wav_ld >= (int64_t)HOP * (p->T_max - 1), "zs_griffin_lim: wav_ld too small");
  hipStream_t main_s = (hipStream_t)stream;
  hipLaunchKernelGGL(gl_tables_kernel, dim3(4), dim3(256), 0, main_s);
  // chains only when every one of them still has a chip's worth of frames per launch and the caller is not capturing a graph
  int chains = g_gl_chains.load(std::memory_order_relaxed);
  while (chains > 1 && ((int64_t)p->n_utt * p->T_max / chains < 8192 || p->n_utt < 2 * chains)) --chains;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (chains > 1 && (hipStreamIsCapturing(main_s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) chains = 1;
  GlSide* g = chains > 1 ? gl_side() : nullptr;
  if (g != nullptr && g->n < chains) chains = g->n;
  g_gl_chains_used.store(g == nullptr || chains < 2 ? 1 : chains, std::memory_order_relaxed);
  if (g == nullptr || chains < 2) return gl_chain(*p, spec_a, spec_b, n_iter, main_s);
  ZS_REQUIRE(hipEventRecord(g->fork, main_s) == hipSuccess, "zs_griffin_lim: event record");
  const int64_t per_utt = (int64_t)p->T_max * NB;
  // every chain gets the tile size the whole batch would get (the default depends on the launch's frame count); the launches are
  // submitted iteration by iteration so that no chain's queue runs dry while another is being filled.
  const int F = p->tile_frames > 0 ? p->tile_frames : ((int64_t)p->n_utt * p->T_max >= 8192 ? GL_TILE_LARGE : GL_TILE_DEFAULT);
  ZsGlIter q[GL_MAX_CHAINS];
  float *cur[GL_MAX_CHAINS], *nxt[GL_MAX_CHAINS];
  hipStream_t cs[GL_MAX_CHAINS];
  // chain boundaries: equal utterance counts, or -- with the lengths on the host -- equal frame counts (the chains then finish
  // together; with 64 utterances of 200..700 frames the thirds differ by up to 10 % in frames)
  int cut[GL_MAX_CHAINS + 1];
  for (int c = 0; c <= chains; ++c) cut[c] = (int)((int64_t)p->n_utt * c / chains);
  if (p->host_lengths != nullptr) {
    int64_t total = 0, run = 0;
    for (int u = 0; u < p->n_utt; ++u) total += p->host_lengths[u];
    int c = 1;
    for (int u = 0; u < p->n_utt && c < chains; ++u) {
      run += p->host_lengths[u];
      if (run * chains >= total * c) cut[c++] = u + 1;
    }
    for (; c < chains; ++c) cut[c] = p->n_utt;
    for (c = 1; c < chains; ++c)                                 // every chain keeps at least one utterance
      if (cut[c] <= cut[c - 1]) cut[c] = cut[c - 1] + 1;
    for (c = chains - 1; c >= 1; --c)
      if (cut[c] >= cut[c + 1]) cut[c] = cut[c + 1] - 1;
  }
  for (int c = 0; c < chains; ++c) {
    const int u0 = cut[c], u1 = cut[c + 1];
    q[c] = *p;
    q[c].mag = p->mag + u0 * per_utt; q[c].lengths = p->lengths + u0; q[c].n_utt = u1 - u0; q[c].wav = p->wav + (int64_t)u0 * p->wav_ld;
    q[c].tile_frames = F;
    cur[c] = spec_a + u0 * per_utt * 2; nxt[c] = spec_b + u0 * per_utt * 2;
    cs[c] = g->s[c];
    ZS_REQUIRE(hipStreamWaitEvent(cs[c], g->fork, 0) == hipSuccess, "zs_griffin_lim: stream wait");
  }
  for (int it = 0; it <= n_iter; ++it)
    for (int c = 0; c < chains; ++c) {
      rc = gl_launch(&q[c], cur[c], it < n_iter ? nxt[c] : nullptr, cs[c], it == 0);
      if (rc) return rc;
      float* t = cur[c]; cur[c] = nxt[c]; nxt[c] = t;
    }
  for (int c = 0; c < chains; ++c)
    ZS_REQUIRE(hipEventRecord(g->join[c], cs[c]) == hipSuccess && hipStreamWaitEvent(main_s, g->join[c], 0) == hipSuccess, "zs_griffin_lim: join");
  return ZS_OK;
}

extern "C" int zs_gl_frame_mse(const float* wav, int64_t wav_ld, const int32_t* lengths, int32_t n_utt, int32_t frame_length, int32_t hop,
                               double* mse, int64_t mse_ld, void* stream) {
  ZS_REQUIRE(wav && lengths && mse && n_utt > 0 && frame_length > 0 && hop > 0 && mse_ld > 0, "zs_gl_frame_mse: bad args");
  const int64_t n_frames = 1 + wav_ld / hop;                                  // upper bound over the utterances (rows hold <= wav_ld samples)
  ZS_REQUIRE(mse_ld >= n_frames, "zs_gl_frame_mse: mse_ld %lld < %lld frames", (long long)mse_ld, (long long)n_frames);
  hipLaunchKernelGGL(gl_frame_mse_kernel, dim3((unsigned)n_frames, (unsigned)n_utt), dim3(256), 0, (hipStream_t)stream, wav, wav_ld, lengths,
                     frame_length, hop, mse, mse_ld);
  return zs_check_launch("zs_gl_frame_mse");
}

extern "C" int zs_gl_deemphasis(float* wav, int64_t wav_ld, const int32_t* lengths, int32_t n_utt, float coef, void* stream) {
  ZS_REQUIRE(wav && lengths && n_utt > 0, "zs_gl_deemphasis: bad args");
  hipLaunchKernelGGL(gl_deemph_scan_kernel, dim3((unsigned)n_utt), dim3(256), 0, (hipStream_t)stream, wav, wav_ld, lengths, coef);
  return zs_check_launch("zs_gl_deemphasis");
}
