// zs_gemm.hip -- MFMA implicit-GEMM Conv1d / Linear (forward + data gradient) and weight gradient.
//
// gfx950 design notes
//  * Four conv-GEMM kernels share one operand layout (channels-last activation rows, weights packed [n][tap][cin_pad],
//    K consumed in 128-byte chunks per row) and one epilogue; zs_gemm_conv picks by problem size:
//      gemm_conv_p8m16_kernel / gemm_conv_p8_kernel   256x256 tile, quadrant ping-pong, LDS-DMA half-tiles (>= 200 tiles)
//      gemm_conv_ring_kernel                            256x128 tile, 3-stage LDS-DMA ring (>= 256 tiles)
//      gemm_conv_dma_kernel                             128x128 tile, LDS-DMA double buffer (default for small layers)
//      gemm_conv_kernel                                 128x128 tile, register-staged (reference variant, ZS_GEMM_DMA=0)
//    and two weight-gradient kernels (gemm_wgrad_p8_kernel bf16 256x256 ping-pong, gemm_wgrad_kernel 128x128).
//  * The MI355X is power-capped (1400 W) in these kernels, not issue-bound: what counts is bytes moved per FLOP and the
//    MFMA shape (16x16x32 is the cheaper instruction), see DESIGN.md section 7.
//  * Operand rows are whole channels-last activation rows, so the conv "im2col" is a per-row pointer
//    computation (tap shift, reflect / zero padding, stride, transposed-conv validity) done by the
//    loader lanes only when the tap changes -- no padded copy of the activation is ever made
//    (the reference does F.pad + conv, model/model.py:36-39).
//  * LDS rows are 128 B with an XOR swizzle applied on the source side of the LDS-DMA (slot s' of row r holds
//    segment s' ^ ((r>>1)&7)); the register-staged kernel pads rows to 144 B instead.  Either way the 16 rows a
//    ds_read_b128 lane group touches fall on 16 distinct 16-byte slots of the 256-byte bank row.
//  * Accumulator layouts: 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); 16x16: col = lane&15,
//    row = 4*(lane>>4) + reg.  Lanes run along the output channel; the epilogue restages through LDS for 16-byte stores.
//  * blockIdx -> tile maps are XCD-aware (tiles that share operand panels get the same XCD's L2).
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "zs_common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int ROWB = 128;    // bytes of K per LDS row
constexpr int PITCH = 144;   // LDS row pitch (bytes)
constexpr int TILE_BYTES = BM * PITCH;

__device__ const uint4 zs_zero_line[8] = {};   // 128 B of zeros: the source of every "zero row" chunk

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// source row of output position t for tap j; ok=false -> the row contributes zeros
__device__ __forceinline__ int conv_src_row(int gather, int pad_mode, int stride, int pad_left, int T_in, int t,
                                            int j, bool& ok) {
  if (gather == 0) {
    int s = t * stride + j - pad_left;
    if (pad_mode == ZS_PAD_REFLECT) s = zs_reflect(s, T_in);
    ok = (s >= 0) && (s < T_in);
    return s;
  }
  int s = t - j;
  ok = false;
  if (s < 0) return 0;
  int q = s / stride;
  ok = (q * stride == s) && (q < T_in);
  return q;
}

// 2-D convolutions (ZsGemmConv.w_in > 0): a sample's rows are an H x w_in image, output position t = (t / w_out, t % w_out), tap
// j = (kw, kh) = (j / taps_h, j % taps_h) -- the 1-D rule applied per axis (same stride, padding and pad mode on both).  The two
// divisions sit in the K loop (every tap change, every row): they are multiplications by 2^32 / d + 1, exact while the dividend
// times the divisor stays below 2^32 (positions per sample x w_out, taps x taps_h: the host checks).
struct Geom2 { int w_in, w_out, taps_h, h_in; unsigned mw, mt; };
__device__ __forceinline__ Geom2 make_geom2(int w_in, int w_out, int taps_h, int T_in) {
  Geom2 g;
  g.w_in = w_in; g.w_out = w_out; g.taps_h = taps_h;
  g.h_in = w_in > 0 ? T_in / w_in : 0;
  g.mw = w_in > 0 ? 0xffffffffu / (unsigned)w_out + 1u : 0u;
  g.mt = w_in > 0 ? 0xffffffffu / (unsigned)taps_h + 1u : 0u;
  return g;
}
__device__ __forceinline__ int conv_src_row_g(int gather, int pad_mode, int stride, int pad_left, int T_in, int t, int j, bool& ok,
                                              const Geom2& g) {
  if (g.w_in == 0) return conv_src_row(gather, pad_mode, stride, pad_left, T_in, t, j, ok);
  const int ho = g.w_out == 1 ? t : (int)__umulhi((unsigned)t, g.mw), wo = t - ho * g.w_out;          // (2^32 / 1 + 1 does not fit)
  const int kw = g.taps_h == 1 ? j : (int)__umulhi((unsigned)j, g.mt), kh = j - kw * g.taps_h;
  bool ok_h, ok_w;
  const int h = conv_src_row(gather, pad_mode, stride, pad_left, g.h_in, ho, kh, ok_h);
  const int w = conv_src_row(gather, pad_mode, stride, pad_left, g.w_in, wo, kw, ok_w);
  ok = ok_h && ok_w;
  return h * g.w_in + w;
}

// (sample, position) of output row m for the A-row loaders.  rb < 0: the row contributes zeros (beyond M, or beyond its sample's
// length in a ragged batch).  With p.lengths the sample's own input length rides in the upper half of rb (see conv_row_*).
__device__ __forceinline__ void conv_row_setup(const ZsGemmConv& p, int m, int M, int& rb, int& rt) {
  if (m >= M) { rb = -1; rt = 0; return; }
  const int b = m / p.T_out;
  rt = m - b * p.T_out;
  rb = b;
  if (p.lengths) {
    const int len = p.lengths[b];
    const int len_out = (len + p.pad_left + p.pad_right - p.taps) / p.stride + 1;
    rb = (rt < len_out) ? ((len << 16) | b) : -1;
  }
}
__device__ __forceinline__ int conv_row_b(const ZsGemmConv& p, int rb) { return p.lengths ? (rb & 0xffff) : rb; }
__device__ __forceinline__ int conv_row_tin(const ZsGemmConv& p, int rb) { return p.lengths ? (rb >> 16) : p.T_in; }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogue.  The accumulators are first transposed through LDS (static register indices, 64 ds_write_b32 per
// lane), then a COMPACT rolled loop finishes the tile with each thread owning 8 consecutive columns of a row:
// 16-byte loads of the mask / residual operands and 16-byte stores (a fully unrolled per-accumulator epilogue
// was ~15k instructions executed once per tile -- instruction-fetch bound, ~22 us per tile -- with 2-byte
// stores).  Order: +bias -> +pre_vec[b] -> act -> *lrelu'(dact_src) -> +add_src -> store out / out2(+vec2[b]).
// ------------------------------------------------------------------------------------------------
constexpr int CPITCH = 132;                       // fp32 row pitch of the staged tile
constexpr int EPI_LDS_BYTES = BM * CPITCH * 4;    // 67,584 B

template <typename T, typename OT>
__device__ __forceinline__ void store_group(OT* base, int64_t row, int64_t ld, int col, int limit, bool aligned,
                                            const float (&v)[8]) {
  OT* d = base + row * ld + col;
  if (aligned && col + 8 <= limit) {
    store8<OT>(d, v);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (col + j < limit) Elem<OT>::st(d + j, v[j]);
  }
}

template <typename T, int NTHREADS, int ROWS = NTHREADS / 2>
__device__ __forceinline__ void epilogue_finish(const ZsGemmConv& p, const float* sC, int M, int m0, int n0, int tid, int g);

// per-sample column sums of the staged [NTHREADS/2 rows][128 columns] fp32 tile (ZsGemmConv.colsum; the host admits it only
// without bias / pre_vec / activation, so the staged accumulators are the values): whole samples per tile, one owner per (b, n)
template <int NTHREADS, int ROWS = NTHREADS / 2>
__device__ __forceinline__ void epilogue_colsum(const ZsGemmConv& p, const float* sC, int M, int m0, int n0, int tid) {
  if (p.colsum == nullptr) return;
  const int ns = ROWS / p.T_out;
  for (int w = tid; w < ns * 128; w += NTHREADS) {
    const int col = w & 127, r0 = (w >> 7) * p.T_out;
    const int n = n0 + col, m = m0 + r0;
    if (n >= p.N || n < p.colsum_col0 || m >= M) continue;
    const float* q = sC + r0 * CPITCH + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = 0;
    if (p.colsum_post && (p.dact_src || p.add_src)) {
      // the stored value: * lrelu'(dact_src) + add_src, element by element (small shapes / fp32 reach this path)
      for (; r < p.T_out; ++r) {
        float v = q[r * CPITCH];
        const int64_t mm = (int64_t)m + r;
        if (p.dact_src) v *= dlrelu_f(p.dtype == ZS_F32 ? ((const float*)p.dact_src)[mm * p.dact_ld + n] : bf2f(((const bf16_t*)p.dact_src)[mm * p.dact_ld + n]), p.slope);
        if (p.add_src) v += (p.add_f32 || p.dtype == ZS_F32) ? ((const float*)p.add_src)[mm * p.add_ld + n] : bf2f(((const bf16_t*)p.add_src)[mm * p.add_ld + n]);
        a0 += v;
      }
    }
    for (; r + 4 <= p.T_out; r += 4) {
      a0 += q[(r + 0) * CPITCH]; a1 += q[(r + 1) * CPITCH]; a2 += q[(r + 2) * CPITCH]; a3 += q[(r + 3) * CPITCH];
    }
    for (; r < p.T_out; ++r) a0 += q[r * CPITCH];
    p.colsum[(int64_t)(m / p.T_out) * p.colsum_ld + (n - p.colsum_col0)] += (a0 + a1) + (a2 + a3);
  }
}

// MI: 32-row MFMA tiles per wave along M (2: the NTHREADS/2-row tile; 1: the 64-row tile of gemm_conv_dma_kernel<T, 64>)
template <typename T, int NTHREADS, int MI = 2>
__device__ __forceinline__ void gemm_epilogue(const ZsGemmConv& p, f32x16 (&acc)[MI][2], float* sC, int M, int m0, int n0,
                                              int wm, int wn, int tid, int g) {
  constexpr int ROWS = (NTHREADS / 2) * MI / 2;
  const int lane = tid & 63;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + ni * 32 + (lane & 31);
        sC[row * CPITCH + col] = acc[mi][ni][r];
      }
  __syncthreads();
  epilogue_colsum<NTHREADS, ROWS>(p, sC, M, m0, n0, tid);
  epilogue_finish<T, NTHREADS, ROWS>(p, sC, M, m0, n0, tid, g);
}

// finishes the staged [ROWS rows][128 columns] fp32 tile at (m0, n0)
template <typename T, int NTHREADS, int ROWS>
__device__ __forceinline__ void epilogue_finish(const ZsGemmConv& p, const float* sC, int M, int m0, int n0, int tid, int g) {
  constexpr int ROWS_PER_PASS = NTHREADS / 16;      // 128-row tile with 256 threads, 256-row tile with 512: 8 passes; 64-row tile: 4
  constexpr int PASSES = ROWS / ROWS_PER_PASS;
  const int cg = tid & 15, rr0 = tid >> 4;          // 16 column groups x 16 rows per pass, 8 passes
  const int n = n0 + cg * 8;
  if (n >= p.N && n >= p.out_cols && n >= p.out2_cols) return;
  const int half = p.N >> 1;
  const bool need_b = (p.pre_vec != nullptr) || (p.vec2 != nullptr);
  const int es = (int)sizeof(T);
  // alignment of the 8-column groups (pointers and pitches are wave-uniform)
  const bool al_out = p.out && ((((uintptr_t)p.out) | (uintptr_t)(p.ldc * (p.out_f32 ? 4 : es)) | (uintptr_t)(p.out_gstride * (p.out_f32 ? 4 : es))) & 15) == 0;
  const bool al_out2 = p.out2 && ((((uintptr_t)p.out2) | (uintptr_t)(p.ldc2 * es)) & 15) == 0;
  const bool al_dact = p.dact_src && ((((uintptr_t)p.dact_src) | (uintptr_t)(p.dact_ld * es)) & 15) == 0;
  const bool al_add = p.add_src && ((((uintptr_t)p.add_src) | (uintptr_t)(p.add_ld * (p.add_f32 ? 4 : es))) & 15) == 0;
  const bool split_ok = (half & 7) == 0;              // an 8-column group never straddles the two pixel-shuffle halves
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = (p.bias != nullptr && n + j < p.N) ? p.bias[(int64_t)g * p.bias_gstride + n + j] : 0.f;
  float* outf = (float*)p.out + (p.out_f32 ? (int64_t)g * p.out_gstride : 0);
  T* outt = (T*)p.out + (p.out_f32 ? 0 : (int64_t)g * p.out_gstride);

#pragma unroll 2
  for (int it = 0; it < PASSES; ++it) {
    const int row = rr0 + ROWS_PER_PASS * it;
    const int m = m0 + row;
    if (m >= M) break;
    float v[8];
    {
      const float4 x0 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8 + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
    int b = 0;
    if (need_b) b = m / p.T_out;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bias[j];
    if (p.pre_vec) {
      const float* pv = p.pre_vec + p.vec_idx[b] * p.pre_vec_ld + n;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (n + j < p.N) v[j] += pv[j];
    }
    if (p.act == ZS_ACT_LRELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = lrelu_f(v[j], p.slope);
    } else if (p.act == ZS_ACT_SIGMOID) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 1.f / (1.f + expf(-v[j]));
    } else if (p.act == ZS_ACT_TANH) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
    }
    if (p.dact_src) {
      const T* dp = (const T*)p.dact_src + (int64_t)m * p.dact_ld + n;
      float y[8];
      if (al_dact && n + 8 <= p.N) load8<T>(dp, y);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (n + j < p.N) ? Elem<T>::ld(dp + j) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= dlrelu_f(y[j], p.slope);
    }
    if (p.add_src) {
      float y[8];
      if (p.add_f32) {
        const float* ap = (const float*)p.add_src + (int64_t)m * p.add_ld + n;
        if (al_add && n + 8 <= p.N) load8<float>(ap, y);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (n + j < p.N) ? ap[j] : 0.f;
        }
      } else {
        const T* ap = (const T*)p.add_src + (int64_t)m * p.add_ld + n;
        if (al_add && n + 8 <= p.N) load8<T>(ap, y);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (n + j < p.N) ? Elem<T>::ld(ap + j) : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += y[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (n + j >= p.N) v[j] = 0.f;      // columns [N, out_cols) are written as zeros

    if (p.out) {
      if (p.store_mode == ZS_STORE_SPLIT2) {
        if (split_ok) {
          const int hi = n >= half;
          if (p.out_f32) store_group<T, float>(outf, 2 * (int64_t)m + hi, p.ldc, n - hi * half, half, al_out, v);
          else store_group<T, T>(outt, 2 * (int64_t)m + hi, p.ldc, n - hi * half, half, al_out, v);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (n + j < p.N) {
              const int hi = (n + j) >= half;
              const int64_t o = (2 * (int64_t)m + hi) * p.ldc + (n + j - hi * half);
              if (p.out_f32) outf[o] = v[j]; else Elem<T>::st(outt + o, v[j]);
            }
        }
      } else {
        if (p.out_f32) store_group<T, float>(outf, m, p.ldc, n, p.out_cols, al_out, v);
        else store_group<T, T>(outt, m, p.ldc, n, p.out_cols, al_out, v);
      }
    }
    if (p.out2) {
      const bool sp = p.store_mode2 == ZS_STORE_SPLIT2;
      if (sp && !split_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (n + j < p.N) {
            const int hi = (n + j) >= half;
            const int oc = n + j - hi * half;
            float w = v[j];
            if (p.vec2) w += p.vec2[p.vec_idx[b] * p.vec2_ld + oc];
            Elem<T>::st((T*)p.out2 + (2 * (int64_t)m + hi) * p.ldc2 + oc, w);
          }
      } else {
        const int hi = sp ? (n >= half) : 0;
        const int oc = n - hi * half;
        const int64_t orow = sp ? (2 * (int64_t)m + hi) : (int64_t)m;
        const int limit = sp ? half : p.out2_cols;
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = v[j];
        if (p.vec2) {
          const float* pv = p.vec2 + p.vec_idx[b] * p.vec2_ld + oc;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < p.N) w[j] += pv[j];
        }
        store_group<T, T>((T*)p.out2, orow, p.ldc2, oc, limit, al_out2, w);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT, 2) void gemm_conv_kernel(const ZsGemmConv p) {
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  constexpr int EPS = 16 / (int)sizeof(T);    // elements per 16-byte segment
  constexpr int KC = ROWB / (int)sizeof(T);   // elements per K chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                    // [2][TILE_BYTES]
  unsigned char* sB = smem + 2 * TILE_BYTES;   // [2][TILE_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.z;
  const T* __restrict__ A = (const T*)p.A + (int64_t)g * p.a_gstride;
  const T* __restrict__ W = (const T*)p.W + (int64_t)g * p.w_gstride;
  const int M = p.B * p.T_out;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (M + BM - 1) / BM;
  // tile order: XCD-contiguous ids, then groups of GM m-tiles swept n-major, so the ~64 tiles resident on one
  // XCD form an (8 x up-to-8) block and share BOTH their A and W panels through that XCD's L2
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int grp = wg / per_group;
  const int gm = min(GM, ntm - grp * GM);           // rows in this (possibly last, shorter) group
  const int in_g = wg - grp * per_group;
  const int m0 = (grp * GM + in_g % gm) * BM, n0 = (in_g / gm) * BN;

  // ---- loader state: 4 rows x one 16-B segment per thread, for A and for W ----
  // cin_pad is a whole number of 128-byte chunks, so a chunk never straddles two taps and the tap change is
  // wave-uniform: per chunk the loader only bumps running pointers; rows that contribute zeros (padding,
  // rows beyond M, invalid transposed-conv positions) point at a resident zero line and do not advance.
  const int seg = tid & 7, r0 = tid >> 3;
  int rb[4], rt[4];            // sample / position of the 4 A rows (rb<0: row beyond M)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    conv_row_setup(p, m, M, rb[i], rt[i]);
  }
  const int chunks_per_tap = p.cin_pad / KC;
  const T* zline = reinterpret_cast<const T*>(zs_zero_line) + seg * EPS;
  const T *pa0, *pa1, *pa2, *pa3;
  int inc0, inc1, inc2, inc3;                 // per-chunk advance (0 for zero rows)
#define ZS_SET_TAP(i, ptr, inc)                                                                         \
  {                                                                                                     \
    bool ok = false; int srow = 0;                                                                      \
    if (rb[i] >= 0) srow = conv_src_row_g(p.gather, p.pad_mode, p.stride, p.pad_left, conv_row_tin(p, rb[i]), rt[i], tap, ok, g2); \
    ptr = ok ? (A + (int64_t)conv_row_b(p, rb[i]) * p.a_batch_stride + (int64_t)srow * p.lda + seg * EPS) : zline;       \
    inc = ok ? KC : 0;                                                                                  \
  }
  const T* pw0 = W + (int64_t)(n0 + r0) * p.ldw + seg * EPS;
  const int64_t wstep = (int64_t)32 * p.ldw;

  const int nk = p.taps * chunks_per_tap;
  // Two register stages of individually named registers, filled/drained by macros: hipcc demoted both arrays and
  // structs captured by (or passed by reference to) the loader lambdas to scratch memory.  Loads run two
  // chunks ahead of the MFMAs.
  uint4 Pa0, Pa1, Pa2, Pa3, Pb0, Pb1, Pb2, Pb3, Qa0, Qa1, Qa2, Qa3, Qb0, Qb1, Qb2, Qb3;
  int tap = 0, cit = 0;                       // wave-uniform K cursor: tap and chunk within tap
  { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
  auto advance = [&]() {
    pw0 += KC;
    if (++cit == chunks_per_tap) {            // uniform branch: next tap -> new source rows
      cit = 0; ++tap;
      if (tap < p.taps) { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
    } else {
      pa0 += inc0; pa1 += inc1; pa2 += inc2; pa3 += inc3;
    }
  };
#define ZS_GLOAD(S)                                                                                      \
  S##a0 = *reinterpret_cast<const uint4*>(pa0); S##a1 = *reinterpret_cast<const uint4*>(pa1);            \
  S##a2 = *reinterpret_cast<const uint4*>(pa2); S##a3 = *reinterpret_cast<const uint4*>(pa3);            \
  S##b0 = *reinterpret_cast<const uint4*>(pw0); S##b1 = *reinterpret_cast<const uint4*>(pw0 + wstep);    \
  S##b2 = *reinterpret_cast<const uint4*>(pw0 + 2 * wstep); S##b3 = *reinterpret_cast<const uint4*>(pw0 + 3 * wstep); \
  advance();
#define ZS_SWRITE(S, buf)                                                                                \
  {                                                                                                      \
    unsigned char* pa_ = sA + (buf) * TILE_BYTES + r0 * PITCH + seg * 16;                                \
    unsigned char* pb_ = sB + (buf) * TILE_BYTES + r0 * PITCH + seg * 16;                                \
    *reinterpret_cast<uint4*>(pa_) = S##a0; *reinterpret_cast<uint4*>(pa_ + 32 * PITCH) = S##a1;          \
    *reinterpret_cast<uint4*>(pa_ + 64 * PITCH) = S##a2; *reinterpret_cast<uint4*>(pa_ + 96 * PITCH) = S##a3; \
    *reinterpret_cast<uint4*>(pb_) = S##b0; *reinterpret_cast<uint4*>(pb_ + 32 * PITCH) = S##b1;          \
    *reinterpret_cast<uint4*>(pb_ + 64 * PITCH) = S##b2; *reinterpret_cast<uint4*>(pb_ + 96 * PITCH) = S##b3; \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* a_base = sA + buf * TILE_BYTES + (wm * 64 + frow) * PITCH + fh * 16;
    const unsigned char* b_base = sB + buf * TILE_BYTES + (wn * 64 + frow) * PITCH + fh * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 fa[2], fb[2];
      fa[0] = *reinterpret_cast<const uint4*>(a_base + ks * 32);
      fa[1] = *reinterpret_cast<const uint4*>(a_base + 32 * PITCH + ks * 32);
      fb[0] = *reinterpret_cast<const uint4*>(b_base + ks * 32);
      fb[1] = *reinterpret_cast<const uint4*>(b_base + 32 * PITCH + ks * 32);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) Mma<T>::run(fa[mi], fb[ni], acc[mi][ni]);
    }
  };

  ZS_GLOAD(P)
  if (nk > 1) { ZS_GLOAD(Q) }
  ZS_SWRITE(P, 0)
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even chunk kt lives in LDS buffer 0; Q holds chunk kt+1 (in flight); P is free
    if (kt + 2 < nk) { ZS_GLOAD(P) }
    compute(0);
    if (kt + 1 < nk) ZS_SWRITE(Q, 1)
    __syncthreads();
    if (kt + 1 >= nk) break;
    // odd chunk kt+1 lives in LDS buffer 1; P holds chunk kt+2 (in flight); Q is free
    if (kt + 3 < nk) { ZS_GLOAD(Q) }
    compute(1);
    if (kt + 2 < nk) ZS_SWRITE(P, 0)
    __syncthreads();
  }
#undef ZS_GLOAD
#undef ZS_SWRITE
#undef ZS_SET_TAP

  gemm_epilogue<T, NT>(p, acc, reinterpret_cast<float*>(smem), M, m0, n0, wm, wn, tid, g);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the same GEMM: operand chunks go global -> LDS directly (global_load_lds_dwordx4, 1 KiB
// per wave-instruction = 8 tile rows x 128 B), no VGPR staging and no ds_write (ds_write_b128 sustains only
// ~79 B/clk/CU, which costs as much LDS-pipe time as the MFMAs of this tile).  The LDS image is lane-linear
// (unpadded 128-byte rows), so bank conflicts are removed by an XOR swizzle applied to the SOURCE segment:
// physical 16-byte slot s' of row r holds logical segment s' ^ ((r >> 1) & 7); fragment reads apply the same
// XOR.  For the 16 rows of a ds_read_b128 lane group the slot index 8*(r&1) + (s ^ ((r>>1)&7)) is a bijection.
// ------------------------------------------------------------------------------------------------
constexpr int DTILE = 128 * 128;   // bytes per operand tile

// BM_ = 128: the tile above.  BM_ = 64: half the rows per workgroup (waves 2 x 2 of 32 x 64) for layers whose 128-row grid leaves
// half the chip idle (the T' = 16 layers: M = 4096, N = 512 -> 128 workgroups of 128 rows, 256 of 64).
template <typename T, int BM_>
__global__ __launch_bounds__(NT, 2) void gemm_conv_dma_kernel(const ZsGemmConv p) {
  constexpr int MI = BM_ / 64, AI = BM_ / 32, ATILE = BM_ * 128;
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  constexpr int EPS = 16 / (int)sizeof(T);
  constexpr int KC = ROWB / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // [2][ATILE]
  unsigned char* sB = smem + 2 * ATILE;     // [2][DTILE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.z;
  const T* __restrict__ A = (const T*)p.A + (int64_t)g * p.a_gstride;
  const T* __restrict__ W = (const T*)p.W + (int64_t)g * p.w_gstride;
  const int M = p.B * p.T_out;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (M + BM_ - 1) / BM_;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int grp = wg / per_group;
  const int gm = min(GM, ntm - grp * GM);
  const int in_g = wg - grp * per_group;
  const int m0 = (grp * GM + in_g % gm) * BM_, n0 = (in_g / gm) * BN;

  // loader: wave-instruction i of this wave fills tile rows wave*32 + 8i .. +7 (lane>>3 selects the row,
  // lane&7 the physical slot); the logical segment it must fetch is slot ^ ((row>>1)&7)
  const int lrow = lane >> 3, slot = lane & 7;
  int rb[4], rt[4], lseg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + lrow;
    lseg[i] = slot ^ ((r >> 1) & 7);               // (rows of the W tile; the A rows of a 64-row tile: see aseg)
    rb[i] = -1; rt[i] = 0;
  }
  int aseg[4];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = wave * (8 * AI) + i * 8 + lrow;
    aseg[i] = slot ^ ((r >> 1) & 7);
    conv_row_setup(p, m0 + r, M, rb[i], rt[i]);
  }
#pragma unroll
  for (int i = AI; i < 4; ++i) aseg[i] = 0;
  const int chunks_per_tap = p.cin_pad / KC;
  const T* zline = reinterpret_cast<const T*>(zs_zero_line);
  const T *pa0, *pa1, *pa2, *pa3;
  int inc0, inc1, inc2, inc3;
#define ZS_SET_TAP(i, ptr, inc)                                                                         \
  {                                                                                                     \
    bool ok = false; int srow = 0;                                                                      \
    if (rb[i] >= 0) srow = conv_src_row_g(p.gather, p.pad_mode, p.stride, p.pad_left, conv_row_tin(p, rb[i]), rt[i], tap, ok, g2); \
    ptr = ok ? (A + (int64_t)conv_row_b(p, rb[i]) * p.a_batch_stride + (int64_t)srow * p.lda + aseg[i] * EPS) : (zline + aseg[i] * EPS); \
    inc = ok ? KC : 0;                                                                                  \
  }
  const T* pw0 = W + (int64_t)(n0 + wave * 32 + lrow) * p.ldw + lseg[0] * EPS;
  const T* pw1 = W + (int64_t)(n0 + wave * 32 + 8 + lrow) * p.ldw + lseg[1] * EPS;
  const T* pw2 = W + (int64_t)(n0 + wave * 32 + 16 + lrow) * p.ldw + lseg[2] * EPS;
  const T* pw3 = W + (int64_t)(n0 + wave * 32 + 24 + lrow) * p.ldw + lseg[3] * EPS;
  const int nk = p.taps * chunks_per_tap;
  int tap = 0, cit = 0;
  pa2 = pa3 = zline; inc2 = inc3 = 0;
  { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) if (AI == 4) { ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) } }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma = [&](int buf) {
    unsigned char* da = sA + buf * ATILE + wave * (1024 * AI);   // this wave's 8 AI rows of A
    unsigned char* db = sB + buf * DTILE + wave * 4096;          // ... and 32 rows of W
    __builtin_amdgcn_global_load_lds((gptr_t)pa0, (lptr_t)(da), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pa1, (lptr_t)(da + 1024), 16, 0, 0);
    if (AI == 4) {
      __builtin_amdgcn_global_load_lds((gptr_t)pa2, (lptr_t)(da + 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)pa3, (lptr_t)(da + 3072), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)pw0, (lptr_t)(db), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pw1, (lptr_t)(db + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pw2, (lptr_t)(db + 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pw3, (lptr_t)(db + 3072), 16, 0, 0);
    pw0 += KC; pw1 += KC; pw2 += KC; pw3 += KC;
    if (++cit == chunks_per_tap) {
      cit = 0; ++tap;
      if (tap < p.taps) { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) if (AI == 4) { ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) } }
    } else {
      pa0 += inc0; pa1 += inc1; pa2 += inc2; pa3 += inc3;
    }
  };
#undef ZS_SET_TAP

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row = w*(32 MI) + mi*32 + (lane&31); logical segment 2*ks + (lane>>5); slot = seg ^ ((row>>1)&7)
  const int frow = lane & 31, fh = lane >> 5;
  const int fx = (frow >> 1) & 7;             // same for row and row+32 (32>>1 = 16, &7 = 0)
  dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // 1) all fragments of chunk kt into registers FIRST: hipcc orders any ds_read that follows an LDS-DMA behind
    //    vmcnt(0) (it cannot tell the two buffers apart), which would serialise the DMA with the MFMAs
    const unsigned char* a_base = sA + buf * ATILE + (wm * (32 * MI) + frow) * 128;
    const unsigned char* b_base = sB + buf * DTILE + (wn * 64 + frow) * 128;
    uint4 fa[4][2], fb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = ((2 * ks + fh) ^ fx) * 16;
      fa[ks][0] = *reinterpret_cast<const uint4*>(a_base + off);
      if (MI == 2) fa[ks][1] = *reinterpret_cast<const uint4*>(a_base + 32 * 128 + off);
      fb[ks][0] = *reinterpret_cast<const uint4*>(b_base + off);
      fb[ks][1] = *reinterpret_cast<const uint4*>(b_base + 32 * 128 + off);
    }
    // 2) chunk kt+1 straight into the other buffer (every wave passed the barrier: nobody reads it any more)
    if (kt + 1 < nk) dma(buf ^ 1);
    // 3) MFMAs run while the DMA is in flight
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) Mma<T>::run(fa[ks][mi], fb[ks][ni], acc[mi][ni]);
    __builtin_amdgcn_sched_barrier(0);                 // keep the MFMAs above the wait (register-only ops float past asm)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __syncthreads();                                   // ... and everyone else's
  }
  gemm_epilogue<T, NT, MI>(p, acc, reinterpret_cast<float*>(smem), M, m0, n0, wm, wn, tid, g);
}

// ------------------------------------------------------------------------------------------------
// Ring variant for problems that fill the chip: 256x128 tile, 512 threads (8 waves as 4x2, each 64x64), one
// workgroup per CU, 3-stage LDS ring (48 KiB per stage: A 256 rows + W 128 rows of 128 B) fed by LDS-DMA with TWO
// chunks (96 KiB per CU) in flight.  Measured on the 128x128 kernel: with the in-loop DMA removed it runs at
// 1290 TFLOP/s, with it at 720 -- the global->LDS stream (bytes in flight x latency), not LDS or MFMA, bounds the
// loop; this variant has 1.5x the bytes in flight and 0.75x the operand bytes per FLOP.
//   iteration k:  s_waitcnt vmcnt(6|0)  (this wave's pieces of chunk k landed; chunk k+1 may still fly)
//                 s_barrier             (everyone's pieces landed AND everyone finished reading stage (k-1)%3)
//                 16 x ds_read_b128     (inline asm: hipcc orders any ds_read behind ALL outstanding LDS-DMA)
//                 DMA chunk k+2 -> stage (k+2)%3 ;  s_waitcnt lgkmcnt(0) ;  16 MFMA
// ------------------------------------------------------------------------------------------------
constexpr int RBM = 256, RNT = 512;
constexpr int RSTAGE = (RBM + BN) * 128;          // 49,152 B
constexpr int RING_LDS = 3 * RSTAGE;              // 147,456 B  (>= epilogue staging 256*132*4 = 135,168 B)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <typename T, int PP>
__global__ __launch_bounds__(RNT, 2) void gemm_conv_ring_kernel(const ZsGemmConv p) {
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  constexpr int EPS = 16 / (int)sizeof(T);
  constexpr int KC = ROWB / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.z;
  const T* __restrict__ A = (const T*)p.A + (int64_t)g * p.a_gstride;
  const T* __restrict__ W = (const T*)p.W + (int64_t)g * p.w_gstride;
  const int M = p.B * p.T_out;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (M + RBM - 1) / RBM;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  constexpr int GM = 4;                              // 4 x 256 rows: the 32 tiles resident on an XCD share panels
  const int per_group = GM * ntn;
  const int grp = wg / per_group;
  const int gm = min(GM, ntm - grp * GM);
  const int in_g = wg - grp * per_group;
  const int m0 = (grp * GM + in_g % gm) * RBM, n0 = (in_g / gm) * BN;

  // loader: per chunk this wave issues 4 A pieces (rows wave*32 + 8i + lane/8) and 2 W pieces (rows wave*16 + 8i + lane/8)
  const int lrow = lane >> 3, slot = lane & 7;
  int rb[4], rt[4], lseg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + lrow;
    lseg[i] = slot ^ ((r >> 1) & 7);
    const int m = m0 + r;
    conv_row_setup(p, m, M, rb[i], rt[i]);
  }
  const int chunks_per_tap = p.cin_pad / KC;
  const T* zline = reinterpret_cast<const T*>(zs_zero_line);
  const T *pa0, *pa1, *pa2, *pa3;
  int inc0, inc1, inc2, inc3;
#define ZS_SET_TAP(i, ptr, inc)                                                                         \
  {                                                                                                     \
    bool ok = false; int srow = 0;                                                                      \
    if (rb[i] >= 0) srow = conv_src_row_g(p.gather, p.pad_mode, p.stride, p.pad_left, conv_row_tin(p, rb[i]), rt[i], tap, ok, g2); \
    ptr = ok ? (A + (int64_t)conv_row_b(p, rb[i]) * p.a_batch_stride + (int64_t)srow * p.lda + lseg[i] * EPS) : (zline + lseg[i] * EPS); \
    inc = ok ? KC : 0;                                                                                  \
  }
  const int wr0 = wave * 16 + lrow, wr1 = wave * 16 + 8 + lrow;
  const T* pw0 = W + (int64_t)(n0 + wr0) * p.ldw + (slot ^ ((wr0 >> 1) & 7)) * EPS;
  const T* pw1 = W + (int64_t)(n0 + wr1) * p.ldw + (slot ^ ((wr1 >> 1) & 7)) * EPS;
  const int nk = p.taps * chunks_per_tap;
  int tap = 0, cit = 0;
  { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // one 1-KiB LDS-DMA piece of the next chunk: i = 0..3 A rows, 4..5 W rows.  Pieces are issued BETWEEN the MFMA groups:
  // an LDS-DMA issued in a phase that also carries ds_read_b128 costs the issuing wave 100-185 cycles, ~60 among MFMAs
  auto dma_piece = [&](int stage, int i) {
    unsigned char* da = smem + stage * RSTAGE + wave * 4096;               // A rows wave*32 ..
    unsigned char* db = smem + stage * RSTAGE + RBM * 128 + wave * 2048;   // W rows wave*16 ..
    if (i == 0) __builtin_amdgcn_global_load_lds((gptr_t)pa0, (lptr_t)(da), 16, 0, 0);
    else if (i == 1) __builtin_amdgcn_global_load_lds((gptr_t)pa1, (lptr_t)(da + 1024), 16, 0, 0);
    else if (i == 2) __builtin_amdgcn_global_load_lds((gptr_t)pa2, (lptr_t)(da + 2048), 16, 0, 0);
    else if (i == 3) __builtin_amdgcn_global_load_lds((gptr_t)pa3, (lptr_t)(da + 3072), 16, 0, 0);
    else if (i == 4) __builtin_amdgcn_global_load_lds((gptr_t)pw0, (lptr_t)(db), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)pw1, (lptr_t)(db + 1024), 16, 0, 0);
  };
  auto dma_advance = [&]() {
    pw0 += KC; pw1 += KC;
    if (++cit == chunks_per_tap) {
      cit = 0; ++tap;
      if (tap < p.taps) { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
    } else {
      pa0 += inc0; pa1 += inc1; pa2 += inc2; pa3 += inc3;
    }
  };
  auto dma = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_piece(stage, i);
    dma_advance();
  };
#undef ZS_SET_TAP

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (LDS byte addresses): row = w*64 + mi*32 + (lane&31); slot = (2ks + lane>>5) ^ ((row>>1)&7)
  const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned a_lane = lds0 + (unsigned)((wm * 64 + frow) * 128);
  const unsigned b_lane = lds0 + (unsigned)(RBM * 128 + (wn * 64 + frow) * 128);
  unsigned koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(((2 * ks + fh) ^ fx) * 16);

  if constexpr (PP) {
    // Ping-pong schedule: waves 0-3 (group 0) and 4-7 (group 1) share the four SIMDs pairwise and run ONE barrier apart, with
    // two barriers per chunk, so that in every barrier interval one group of a SIMD reads fragments (L) while the other runs
    // its 16 MFMAs (M):
    //   group 0:      L0 | M0 | L1 | M1 | ...        L_k: 16 ds_read(chunk k) [+ g0: DMA chunk k+2]; lgkmcnt(0); [g1: vmcnt]
    //   group 1:   -  |  L0 | M0 | L1 | M1 ...       M_k: 16 MFMA [+ g1: DMA chunk k+3 between them]; [g0: vmcnt]
    // A chunk is read only after every wave waited for its own DMA pieces of it and a barrier followed (RAW); a stage is
    // refilled only after both groups' lgkmcnt(0) of the chunk it held and a barrier (WAR).
    const int grp1 = wave >> 2;
    dma(0);
    if (nk > 1) dma(1);
    if (grp1 && nk > 2) dma(2);
    {
      const int fly = (grp1 ? min(nk, 3) : min(nk, 2)) - 1;      // chunks that may still be in flight
      if (fly == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (fly == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp1) __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned sa = a_lane + (unsigned)(stage * RSTAGE), sb = b_lane + (unsigned)(stage * RSTAGE);
      u32x4_t fa[4][2], fb[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][0]) : "v"(sa + koff[ks]));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[ks][1]) : "v"(sa + koff[ks]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[ks][0]) : "v"(sb + koff[ks]));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[ks][1]) : "v"(sb + koff[ks]));
      }
      int s2 = stage + 2; if (s2 >= 3) s2 -= 3;
      if (!grp1 && kt + 2 < nk) dma(s2);                          // group 0: chunk kt+2 -> stage (kt+2)%3
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (grp1) {                                                 // group 1: its pieces of chunk kt+1 have landed
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool more1 = grp1 && (kt + 3 < nk);                   // group 1: chunk kt+3 -> stage kt%3 (read by everyone already)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const uint4 ua = make_uint4(fa[ks][mi].x, fa[ks][mi].y, fa[ks][mi].z, fa[ks][mi].w);
            const uint4 ub = make_uint4(fb[ks][ni].x, fb[ks][ni].y, fb[ks][ni].z, fb[ks][ni].w);
            Mma<T>::run(ua, ub, acc[mi][ni]);
          }
        if (more1 && ks < 3) {
          __builtin_amdgcn_sched_barrier(0);
          dma_piece(stage, 2 * ks);
          dma_piece(stage, 2 * ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (more1) dma_advance();
      if (!grp1) {                                                // group 0: its pieces of chunk kt+1 have landed
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (++stage == 3) stage = 0;
    }
    if (!grp1) __builtin_amdgcn_s_barrier();                      // pairs with group 1's last barrier
  } else {
  dma(0);
  if (nk > 1) dma(1);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // chunk kt landed, chunk kt+1 (6 pieces) may fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned sa = a_lane + (unsigned)(stage * RSTAGE), sb = b_lane + (unsigned)(stage * RSTAGE);
    u32x4_t fa[4][2], fb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][0]) : "v"(sa + koff[ks]));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[ks][1]) : "v"(sa + koff[ks]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[ks][0]) : "v"(sb + koff[ks]));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[ks][1]) : "v"(sb + koff[ks]));
    }
    const bool more = kt + 2 < nk;                 // chunk kt+2 -> stage (kt+2)%3 == (kt-1)%3: nobody reads it any more
    int s2 = stage + 2; if (s2 >= 3) s2 -= 3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);            // MFMAs (register-only) must stay below the wait
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const uint4 ua = make_uint4(fa[ks][mi].x, fa[ks][mi].y, fa[ks][mi].z, fa[ks][mi].w);
          const uint4 ub = make_uint4(fb[ks][ni].x, fb[ks][ni].y, fb[ks][ni].z, fb[ks][ni].w);
          Mma<T>::run(ua, ub, acc[mi][ni]);
        }
      if (more && ks < 3) {                        // two DMA pieces behind each of the first three MFMA groups
        __builtin_amdgcn_sched_barrier(0);
        dma_piece(s2, 2 * ks);
        dma_piece(s2, 2 * ks + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) dma_advance();
    __builtin_amdgcn_sched_barrier(0);
    if (++stage == 3) stage = 0;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                // all fragment reads done before the ring is reused as the staging tile
  gemm_epilogue<T, RNT>(p, acc, reinterpret_cast<float*>(smem), M, m0, n0, wm, wn, tid, g);
}

// ------------------------------------------------------------------------------------------------
// 256x256 "quadrant ping-pong" kernel for the large layers.  A power-capped MI355X (1400 W: the 128x128 / 256x128 kernels
// above already draw 1350 W at 0.9 PFLOP/s, sclk 2.03 GHz) is faster only with fewer bytes moved per FLOP, so:
//   * 256x256 tile, 8 waves as 2(M) x 4(N), each 128x64 = 2x2 quadrants of 64x32 (two 32x32x16 MFMA tiles x 4 k-steps):
//     1/128 operand byte from L2 per FLOP (256x128: 3/256) and 3/128 fragment bytes from LDS per FLOP per wave (64x64: 4/128).
//   * K tile = 128 B per row, 64 KiB per stage (A 256 rows | W 256 rows), TWO stages, each split in four 16-KiB half-tiles
//     (A0 A1 B0 B1) that are refilled by LDS-DMA as soon as their last reader is done:
//        during tile k (stage s):  phase 1: A0(k+1) -> s^1   phase 2: A1(k+1) -> s^1   phase 3: B0(k+2) -> s   phase 4: B1(k+2) -> s
//     (every wave issues 2 pieces per phase, between its MFMAs), so 3 half-tiles stay in flight across the barriers.
//   * Four phases per K tile, one quadrant each:  (0,0) reads A[mq0] (8 ds_read_b128) + B[nq0] (4) | (0,1) reads B[nq1] (4) |
//     (1,1) reads A[mq1] (8) | (1,0) reads nothing.  Phase = L (fragment reads, lgkmcnt(0)) ; s_barrier ; M (8 MFMA + 2 DMA
//     pieces) ; s_barrier.  Waves 0-3 (M half 0) and 4-7 (M half 1) share the SIMDs pairwise and run ONE barrier apart, so on
//     every SIMD one wave is in M while the other is in L.
//   * One counted wait per K tile: group 0 at the end of M4 (vmcnt 4: only B0/B1(k+2) may fly), group 1 at the end of its L4
//     (vmcnt 2: it has not issued B1(k+2) yet); both sit before the barrier that precedes the first read of tile k+1.
//     WAR: a half-tile is refilled at least one barrier after the lgkmcnt(0) of its last reader (A halves are read by one
//     group only: phases 1 and 3; B halves by both: phases 1 and 2).
// ------------------------------------------------------------------------------------------------
constexpr int PBM = 256, PBN = 256, PNT = 512;
constexpr int PSTAGE = (PBM + PBN) * 128;                     // 65,536 B
constexpr int P8_EPI = 256 * CPITCH * 4;                      // 135,168 B: staging of a 256 x 128 fp32 half
constexpr int P8_LDS = (2 * PSTAGE > P8_EPI) ? 2 * PSTAGE : P8_EPI;

#define ZS_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define ZS_P8_DMA(cond, src, dst)                                                \
  __builtin_amdgcn_sched_barrier(0);                                             \
  if (cond) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
// ---- MFMA shape: v_mfma_f32_16x16x32_bf16 (fp32: 16x16x4).  A pure-MFMA probe (tools/mfma_power.hip) sustains 2.07 PFLOP/s
// with the 16x16x32 shape against 1.80 PFLOP/s with 32x32x16 on random operands under the 1400 W cap, i.e. the smaller tile is
// the more energy-efficient instruction (a 32x32x16 twin of this kernel was measured and removed in round 3).  A quadrant is
// 4x2 tiles of 16x16 x 2 k-steps of 32 = 16 MFMAs.
typedef __attribute__((ext_vector_type(4))) float f32x4_m;
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_m& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_m& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
#define ZS_M16_READ_A(o0, o1, o2, o3)                                            \
  ZS_DSR(fa00, sa + koff0, o0); ZS_DSR(fa01, sa + koff1, o0); ZS_DSR(fa10, sa + koff0, o1); ZS_DSR(fa11, sa + koff1, o1); \
  ZS_DSR(fa20, sa + koff0, o2); ZS_DSR(fa21, sa + koff1, o2); ZS_DSR(fa30, sa + koff0, o3); ZS_DSR(fa31, sa + koff1, o3);
#define ZS_M16_READ_B(f0, f1, f2, f3, o0, o1)                                    \
  ZS_DSR(f0, sb + koff0, o0); ZS_DSR(f1, sb + koff1, o0); ZS_DSR(f2, sb + koff0, o1); ZS_DSR(f3, sb + koff1, o1);
#define ZS_M16_MMA(a, b, cc) Mma16<T>::run(make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w), cc)
// b0/b1 = n-tile 0 at k-step 0/1, b2/b3 = n-tile 1 at k-step 0/1; cq = c[mq][nq] ([mt][nt])
#define ZS_M16_QUAD(cq, b0, b1, b2, b3, cond, src0, dst0, src1, dst1)            \
  __builtin_amdgcn_s_setprio(1);                                                 \
  ZS_M16_MMA(fa00, b0, cq[0][0]); ZS_M16_MMA(fa00, b2, cq[0][1]); ZS_M16_MMA(fa10, b0, cq[1][0]); ZS_M16_MMA(fa10, b2, cq[1][1]); \
  ZS_P8_DMA(cond, src0, dst0)                                                    \
  ZS_M16_MMA(fa20, b0, cq[2][0]); ZS_M16_MMA(fa20, b2, cq[2][1]); ZS_M16_MMA(fa30, b0, cq[3][0]); ZS_M16_MMA(fa30, b2, cq[3][1]); \
  ZS_M16_MMA(fa01, b1, cq[0][0]); ZS_M16_MMA(fa01, b3, cq[0][1]);                \
  ZS_P8_DMA(cond, src1, dst1)                                                    \
  ZS_M16_MMA(fa11, b1, cq[1][0]); ZS_M16_MMA(fa11, b3, cq[1][1]); ZS_M16_MMA(fa21, b1, cq[2][0]); ZS_M16_MMA(fa21, b3, cq[2][1]); \
  ZS_M16_MMA(fa31, b1, cq[3][0]); ZS_M16_MMA(fa31, b3, cq[3][1]);                \
  __builtin_amdgcn_s_setprio(0);

__device__ __forceinline__ bool al16(const void* q, int64_t pitch_bytes) { return ((((uintptr_t)q) | (uintptr_t)pitch_bytes) & 15) == 0; }

// which problems the register epilogue of the 256x256 kernel takes (wave-uniform: depends on the launch parameters only)
__device__ __forceinline__ bool p8_regs_epilogue_ok(const ZsGemmConv& p) {
  const int half = p.N >> 1;
  const bool split = (p.out && p.store_mode == ZS_STORE_SPLIT2) || (p.out2 && p.store_mode2 == ZS_STORE_SPLIT2);
  return (p.out == nullptr || (!p.out_f32 && al16(p.out, p.ldc * 2) && ((p.out_gstride * 2) & 15) == 0 && (p.out_cols & 7) == 0)) &&
         (p.out2 == nullptr || (al16(p.out2, p.ldc2 * 2) && (p.out2_cols & 7) == 0 && p.dact_src == nullptr && p.add_src == nullptr)) &&
         (p.dact_src == nullptr || al16(p.dact_src, p.dact_ld * 2)) &&
         (p.add_src == nullptr || (!p.add_f32 && al16(p.add_src, p.add_ld * 2))) &&
         ((p.pre_vec == nullptr && p.vec2 == nullptr) || (p.T_out & 15) == 0) && (!split || (half & 7) == 0) &&
         (p.act == ZS_ACT_NONE || p.act == ZS_ACT_LRELU) &&
         (p.colsum == nullptr || ((PBM % p.T_out) == 0 && p.out2 == nullptr));
}

// Register epilogue of the 256x256 kernels (bf16 outputs): bias, the per-sample vector and the activation are applied to the
// accumulators in registers and the WHOLE tile is staged as bf16 (132 KiB), so the finish is a deeply unrolled LDS -> global copy
// with 16-byte accesses that also applies what needs a second operand (lrelu' mask of dact_src, add_src; all of a thread's loads
// in flight at once) and the pixel-shuffle row mapping.  out2 (= value + vec2[idx[b]]) is staged and copied the same way from
// the values still in registers; colsum (per-sample column sums: the nn.Embedding part of the backward) reads the staged tile.
// HAS2 (out2 given) is a template parameter so that without it the accumulators die at the first staging (registers for the copy).
template <bool HAS2>
__device__ __forceinline__ void p8_regs_epilogue(const ZsGemmConv& p, f32x4_m (&c)[2][2][4][2], unsigned char* smem, int M, int m0, int n0,
                                                 int tid, int lane, int wr, int wc, int g) {
  constexpr int PT = 264;                                           // bf16 pitch: 528-byte rows
  unsigned short* sT = reinterpret_cast<unsigned short*>(smem);
  const int colb = wc * 64 + (lane & 15);
  const int rbase = wr * 128 + 4 * (lane >> 4);
  const int half = p.N >> 1;
  // sample (row of pre_vec / vec2) of each 16-row block of this wave: T_out is a multiple of 16 when they are given
  int64_t bsel[2][4];
  const bool need_b = p.pre_vec != nullptr || p.vec2 != nullptr;
#pragma unroll
  for (int mq = 0; mq < 2; ++mq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int mb = m0 + wr * 128 + mq * 64 + mt * 16;
      bsel[mq][mt] = (need_b && mb < M) ? p.vec_idx[mb / p.T_out] : 0;
    }
#pragma unroll
  for (int nq = 0; nq < 2; ++nq)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = colb + nq * 32 + nt * 16;
      const int n = n0 + col;
      const bool cval = n < p.N;
      const float bv = (p.bias != nullptr && cval) ? p.bias[(int64_t)g * p.bias_gstride + n] : 0.f;
      float pv[2][4];
#pragma unroll
      for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) pv[mq][mt] = bv;
      if (p.pre_vec != nullptr && cval) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) pv[mq][mt] += p.pre_vec[bsel[mq][mt] * p.pre_vec_ld + n];
      }
#pragma unroll
      for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            float v = c[mq][nq][mt][nt][i4] + pv[mq][mt];
            if (p.act == ZS_ACT_LRELU) v = lrelu_f(v, p.slope);
            if (!cval) v = 0.f;                                     // columns [N, out_cols) are written as zeros
            if constexpr (HAS2) c[mq][nq][mt][nt][i4] = v;
            sT[(rbase + mq * 64 + mt * 16 + i4) * PT + col] = f2bf(v);
          }
    }
  __syncthreads();

  const bool cs_post = !HAS2 && p.colsum != nullptr && p.colsum_post && (p.dact_src != nullptr || p.add_src != nullptr);
  auto colsum_pass = [&]() {
    // per-sample column sums of the staged values: one thread per (column, half of the tile's samples);
    // a tile holds whole samples (256 % T_out == 0), so every (sample, column) has one owner: plain += , no atomics
    const int ns = PBM / p.T_out;
    const int col = tid & 255, n = n0 + col, hs = tid >> 8;
    const int s0 = ns >= 2 ? hs * (ns >> 1) : 0, s1 = ns >= 2 ? s0 + (ns >> 1) : (hs == 0 ? 1 : 0);
    if (n < p.N && n >= p.colsum_col0) {
      for (int s = s0; s < s1; ++s) {
        const int r0 = s * p.T_out, m = m0 + r0;
        if (m >= M) break;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const unsigned short* q = sT + r0 * PT + col;
        int r = 0;
        for (; r + 4 <= p.T_out; r += 4) {
          a0 += bf2f(q[(r + 0) * PT]); a1 += bf2f(q[(r + 1) * PT]); a2 += bf2f(q[(r + 2) * PT]); a3 += bf2f(q[(r + 3) * PT]);
        }
        for (; r < p.T_out; ++r) a0 += bf2f(q[r * PT]);
        float* d = p.colsum + (int64_t)(m / p.T_out) * p.colsum_ld + (n - p.colsum_col0);
        *d += (a0 + a1) + (a2 + a3);
      }
    }
  };
  if (!HAS2 && p.colsum != nullptr && !cs_post) colsum_pass();      // of the value before mask / add

  const int c8 = (tid & 31) * 8, r0 = tid >> 5;                     // 32 x 16-byte groups per row, 16 rows per pass
  const int n = n0 + c8;
  if (p.out != nullptr) {
    unsigned short* outp = (unsigned short*)p.out + (int64_t)g * p.out_gstride;
    const bool sp = p.store_mode == ZS_STORE_SPLIT2;
    const int hi = sp ? (n >= half) : 0;
    const int oc = n - hi * half;
    const bool cok = sp ? (n < p.N) : (n < p.out_cols);
    const bool second = p.dact_src != nullptr || p.add_src != nullptr;
    if (cok && !second) {
#pragma unroll 8
      for (int it = 0; it < 16; ++it) {
        const int row = r0 + 16 * it;
        const int m = m0 + row;
        if (m < M) {
          const uint4 w = *reinterpret_cast<const uint4*>(sT + row * PT + c8);
          const int64_t orow = sp ? 2 * (int64_t)m + hi : (int64_t)m;
          *reinterpret_cast<uint4*>(outp + orow * p.ldc + oc) = w;
        }
      }
    } else if (!HAS2 && cok) {                                      // (out2 excludes dact_src / add_src: p8_regs_epilogue_ok)
      const bf16_t* dsrc = (const bf16_t*)p.dact_src;
      const bf16_t* asrc = (const bf16_t*)p.add_src;
      const bool tail = n + 8 > p.N;                               // the group reaches into the zero columns [N, out_cols)
#pragma unroll 1
      for (int it0 = 0; it0 < 16; it0 += 8) {
        Raw8<bf16_t> rd[8], ra[8];
        uint4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = r0 + 16 * (it0 + u);
          const int m = min(m0 + row, M - 1);
          if (dsrc) rd[u].ld(dsrc + (int64_t)m * p.dact_ld + n);
          if (asrc) ra[u].ld(asrc + (int64_t)m * p.add_ld + n);
          w[u] = *reinterpret_cast<const uint4*>(sT + row * PT + c8);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = r0 + 16 * (it0 + u);
          const int m = m0 + row;
          if (m < M) {
            float v[8], y[8];
            Raw8<bf16_t> rw; rw.a = w[u]; rw.cvt(v);
            if (dsrc) {
              rd[u].cvt(y);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= dlrelu_f(y[j], p.slope);
            }
            if (asrc) {
              ra[u].cvt(y);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += y[j];
            }
            if (tail) {
#pragma unroll
              for (int j = 0; j < 8; ++j) if (n + j >= p.N) v[j] = 0.f;
            }
            const int64_t orow = sp ? 2 * (int64_t)m + hi : (int64_t)m;
            store8<bf16_t>(outp + orow * p.ldc + oc, v);
            if (cs_post) store8<bf16_t>(sT + row * PT + c8, v);     // the stored value back into the staging tile (own slot)
          }
        }
      }
    }
  }
  if (cs_post) {                                                    // column sums of the STORED values (colsum_post)
    __syncthreads();
    colsum_pass();
  }
  if constexpr (HAS2) {
    __syncthreads();                                                // every thread is done with the first staging
    const bool sp = p.store_mode2 == ZS_STORE_SPLIT2;
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = colb + nq * 32 + nt * 16;
        const int nn = n0 + col;
        const bool cval = nn < p.N;
        const int ocv = sp ? (nn >= half ? nn - half : nn) : nn;
        float ev[2][4];
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) ev[mq][mt] = 0.f;
        if (p.vec2 != nullptr && cval) {
#pragma unroll
          for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) ev[mq][mt] = p.vec2[bsel[mq][mt] * p.vec2_ld + ocv];
        }
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
              sT[(rbase + mq * 64 + mt * 16 + i4) * PT + col] = f2bf(c[mq][nq][mt][nt][i4] + ev[mq][mt]);
      }
    __syncthreads();
    unsigned short* outp = (unsigned short*)p.out2;
    const int hi = sp ? (n >= half) : 0;
    const int oc = n - hi * half;
    const bool cok = sp ? (n < p.N) : (n < p.out2_cols);
    if (cok) {
#pragma unroll 8
      for (int it = 0; it < 16; ++it) {
        const int row = r0 + 16 * it;
        const int m = m0 + row;
        if (m < M) {
          const uint4 w = *reinterpret_cast<const uint4*>(sT + row * PT + c8);
          const int64_t orow = sp ? 2 * (int64_t)m + hi : (int64_t)m;
          *reinterpret_cast<uint4*>(outp + orow * p.ldc2 + oc) = w;
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(PNT, 2) void gemm_conv_p8m16_kernel(const ZsGemmConv p) {
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  constexpr int EPS = 16 / (int)sizeof(T);
  constexpr int KC = ROWB / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = blockIdx.z;
  const T* __restrict__ A = (const T*)p.A + (int64_t)g * p.a_gstride;
  const T* __restrict__ W = (const T*)p.W + (int64_t)g * p.w_gstride;
  const int M = p.B * p.T_out;
  const int ntn = (p.N + PBN - 1) / PBN;
  const int ntm = (M + PBM - 1) / PBM;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int grp = wg / per_group;
  const int gm = min(GM, ntm - grp * GM);
  const int in_g = wg - grp * per_group;
  const int m0 = (grp * GM + in_g % gm) * PBM, n0 = (in_g / gm) * PBN;

  // loader: piece j = 0..3 of a K tile covers rows (j>>1)*128 + wave*16 + (j&1)*8 + lane/8 of A and of W
  const int lrow = lane >> 3, slot = lane & 7;
  int rb[4], rt[4], lseg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (j >> 1) * 128 + wave * 16 + (j & 1) * 8 + lrow;
    lseg[j] = slot ^ ((r >> 1) & 7);
    const int m = m0 + r;
    conv_row_setup(p, m, M, rb[j], rt[j]);
  }
  const int chunks_per_tap = p.cin_pad / KC;
  const T* zline = reinterpret_cast<const T*>(zs_zero_line);
  const T *pa0, *pa1, *pa2, *pa3;
  int inc0, inc1, inc2, inc3;
#define ZS_SET_TAP(i, ptr, inc)                                                                         \
  {                                                                                                     \
    bool ok = false; int srow = 0;                                                                      \
    if (rb[i] >= 0) srow = conv_src_row_g(p.gather, p.pad_mode, p.stride, p.pad_left, conv_row_tin(p, rb[i]), rt[i], tap, ok, g2); \
    ptr = ok ? (A + (int64_t)conv_row_b(p, rb[i]) * p.a_batch_stride + (int64_t)srow * p.lda + lseg[i] * EPS) : (zline + lseg[i] * EPS); \
    inc = ok ? KC : 0;                                                                                  \
  }
  const int nk = p.taps * chunks_per_tap;
  int tap = 0, cit = 0;
  { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
  auto advance_a = [&]() {
    if (++cit == chunks_per_tap) {
      cit = 0; ++tap;
      if (tap < p.taps) { ZS_SET_TAP(0, pa0, inc0) ZS_SET_TAP(1, pa1, inc1) ZS_SET_TAP(2, pa2, inc2) ZS_SET_TAP(3, pa3, inc3) }
    } else {
      pa0 += inc0; pa1 += inc1; pa2 += inc2; pa3 += inc3;
    }
  };
#undef ZS_SET_TAP
  // W rows n0 + r_j (the packed matrix has n_pad >= n0 + 256 rows: checked on the host)
  const T* pw0 = W + (int64_t)(n0 + wave * 16 + lrow) * p.ldw + lseg[0] * EPS;
  const T* pw1 = W + (int64_t)(n0 + wave * 16 + 8 + lrow) * p.ldw + lseg[1] * EPS;
  const int64_t whalf = (int64_t)128 * p.ldw;                 // rows +128: pieces 2, 3

  unsigned char* const ldsA = smem + wave * 2048;             // + stage*PSTAGE + (j>>1)*16384 + (j&1)*1024
  unsigned char* const ldsB = smem + 32768 + wave * 2048;

  f32x4_m c[2][2][4][2];                                      // c[mq][nq][mt][nt]: 16x16 tiles
#pragma unroll
  for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
    for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int i3 = 0; i3 < 2; ++i3) c[i0][i1][i2][i3] = f32x4_m{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (16x16x32 MFMA): A row = wr*128 + mq*64 + mt*16 + (lane&15), W row = wc*64 + nq*32 + nt*16 + (lane&15);
  // slot = (4ks + lane>>4) ^ ((row>>1)&7)
  const int frow = lane & 15, fh = lane >> 4, fx = (frow >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned a_lane = lds0 + (unsigned)((wr * 128 + frow) * 128);
  const unsigned b_lane = lds0 + (unsigned)(32768 + (wc * 64 + frow) * 128);
  const unsigned koff0 = (unsigned)(((0 + fh) ^ fx) * 16), koff1 = (unsigned)(((4 + fh) ^ fx) * 16);
  const int grp1 = wr;

  // prologue: A(0) B(0) -> stage 0, B(1) -> stage 1
  __builtin_amdgcn_global_load_lds((gptr_t)pa0, (lptr_t)(ldsA), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)pa1, (lptr_t)(ldsA + 1024), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)pa2, (lptr_t)(ldsA + 16384), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)pa3, (lptr_t)(ldsA + 16384 + 1024), 16, 0, 0);
  advance_a();
  __builtin_amdgcn_global_load_lds((gptr_t)pw0, (lptr_t)(ldsB), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)pw1, (lptr_t)(ldsB + 1024), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)(pw0 + whalf), (lptr_t)(ldsB + 16384), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)(pw1 + whalf), (lptr_t)(ldsB + 16384 + 1024), 16, 0, 0);
  pw0 += KC; pw1 += KC;
  if (nk > 1) {
    __builtin_amdgcn_global_load_lds((gptr_t)pw0, (lptr_t)(ldsB + PSTAGE), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pw1, (lptr_t)(ldsB + PSTAGE + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(pw0 + whalf), (lptr_t)(ldsB + PSTAGE + 16384), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(pw1 + whalf), (lptr_t)(ldsB + PSTAGE + 16384 + 1024), 16, 0, 0);
    pw0 += KC; pw1 += KC;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp1) __builtin_amdgcn_s_barrier();

  u32x4_t fa00, fa01, fa10, fa11, fa20, fa21, fa30, fa31;    // fa[mt][ks]
  u32x4_t fb00, fb01, fb02, fb03, fb10, fb11, fb12, fb13;    // fb[nq][2*nt + ks]
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    const unsigned sa = a_lane + (unsigned)(st * PSTAGE), sb = b_lane + (unsigned)(st * PSTAGE);
    const bool has_a = kt + 1 < nk, has_b = kt + 2 < nk;
    unsigned char* const dA = ldsA + (st ^ 1) * PSTAGE;
    unsigned char* const dB = ldsB + st * PSTAGE;
    // ---- phase 1: quadrant (0,0); A0(kt+1)
    ZS_M16_READ_A(0, 2048, 4096, 6144)
    ZS_M16_READ_B(fb00, fb01, fb02, fb03, 0, 2048)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_M16_QUAD(c[0][0], fb00, fb01, fb02, fb03, has_a, pa0, dA, pa1, dA + 1024)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: quadrant (0,1); A1(kt+1)
    ZS_M16_READ_B(fb10, fb11, fb12, fb13, 4096, 6144)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_M16_QUAD(c[0][1], fb10, fb11, fb12, fb13, has_a, pa2, dA + 16384, pa3, dA + 16384 + 1024)
    if (has_a) advance_a();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: quadrant (1,1); B0(kt+2)
    ZS_M16_READ_A(8192, 10240, 12288, 14336)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_M16_QUAD(c[1][1], fb10, fb11, fb12, fb13, has_b, pw0, dB, pw1, dB + 1024)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 4: quadrant (1,0); B1(kt+2); the counted waits for tile kt+1
    if (grp1) {
      if (has_b) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_M16_QUAD(c[1][0], fb00, fb01, fb02, fb03, has_b, pw0 + whalf, dB + 16384, pw1 + whalf, dB + 16384 + 1024)
    if (has_b) { pw0 += KC; pw1 += KC; }
    if (!grp1) {
      if (has_b) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!grp1) __builtin_amdgcn_s_barrier();          // pairs with group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // register epilogue (bf16 outputs; see p8_regs_epilogue): the two-half fp32 path below keeps two rows in flight per thread and
  // costs as much as the whole K loop of a 1024-deep layer -- it is left for fp32 outputs and odd alignments
  if constexpr (sizeof(T) == 2) {
    if (p8_regs_epilogue_ok(p)) {
      if (p.out2 != nullptr) p8_regs_epilogue<true>(p, c, smem, M, m0, n0, tid, lane, wr, wc, g);
      else p8_regs_epilogue<false>(p, c, smem, M, m0, n0, tid, lane, wr, wc, g);
      return;
    }
  }
  // epilogue in two 128-column halves through the [256][132] fp32 staging tile
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if ((wc >> 1) == h) {
      const int cb = (wc & 1) * 64 + (lane & 15);
      const int rbase = wr * 128 + 4 * (lane >> 4);
#pragma unroll
      for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                sC[(rbase + mq * 64 + mt * 16 + i) * CPITCH + cb + nq * 32 + nt * 16] = c[mq][nq][mt][nt][i];
    }
    __syncthreads();
    epilogue_colsum<PNT>(p, sC, M, m0, n0 + h * 128, tid);
    epilogue_finish<T, PNT>(p, sC, M, m0, n0 + h * 128, tid, g);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: slab[split][co][tap][ci] = sum_{m in split} dY[m][co] * X[row(m,tap)][ci]
// Both operands are K(=row)-strided in memory.  fp32 fragments are natural (lane = channel, one
// ds_read_b32 per MFMA operand); bf16 fragments gather 8 rows per lane with ds_read_u16.
// ------------------------------------------------------------------------------------------------
constexpr int WK = 32;   // rows (K) per chunk

template <typename T> struct WFrag;
template <> struct WFrag<float> {
  static constexpr int PITCHW = 128 * 4 + 16;
  // chunk of 32 rows -> 16 steps of v_mfma_f32_32x32x2_f32 (k = 2*step + lane>>5)
  static __device__ __forceinline__ void chunk(const unsigned char* sY, const unsigned char* sX, int wm, int wn, int lane,
                                               f32x16 (&acc)[2][2]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float*>(sY + (2 * st + h) * PITCHW + (wm * 64 + i * 32 + r) * 4);
        b[i] = *reinterpret_cast<const float*>(sX + (2 * st + h) * PITCHW + (wn * 64 + i * 32 + r) * 4);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
};
template <> struct WFrag<bf16_t> {
  // 320-byte rows: for ds_read_b64_tr_b16 the four row addresses of a 16-lane group start 80 dwords apart
  // (0,16,32,48 mod 64) and the second group of a 32-lane half is 8 dwords further: conflict free.
  static constexpr int PITCHW = 128 * 2 + 64;
  // chunk of 32 rows -> 2 steps of v_mfma_f32_32x32x16_bf16.  Both operands are K(=row)-strided in LDS, so the
  // fragments come from the hardware transposing read: per 16-lane group, lane 4q+p supplies the address of row
  // q, columns 4p..4p+3 of a 4-row x 16-column block and lane i receives column i of those 4 rows.  A lane's
  // 8 k-values (k = 8*(lane>>5) + j) are two such reads (rows kb..kb+3, kb+4..kb+7); 2 LDS instructions per
  // fragment instead of 8 ds_read_u16.  EXEC is all ones here (uniform control flow, 256-thread blocks).
  static __device__ __forceinline__ void chunk(const unsigned char* sY, const unsigned char* sX, int wm, int wn, int lane,
                                               f32x16 (&acc)[2][2]) {
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int cb = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const unsigned ya = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)sY +
                        (unsigned)((kb + q) * PITCHW + (wm * 64 + cb + 4 * pp) * 2);
    const unsigned xa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)sX +
                        (unsigned)((kb + q) * PITCHW + (wn * 64 + cb + 4 * pp) * 2);
    uint2 ra[2][2][2], rb[2][2][2];            // [k-step][32-wide tile][row half]
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const unsigned off = (unsigned)((16 * st + 4 * hb) * PITCHW + t * 64);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ra[st][t][hb]) : "v"(ya + off));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rb[st][t][hb]) : "v"(xa + off));
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);         // MFMAs (register-only) must not be hoisted above the wait
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      uint4 fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t] = make_uint4(ra[st][t][0].x, ra[st][t][0].y, ra[st][t][1].x, ra[st][t][1].y);
        fb[t] = make_uint4(rb[st][t][0].x, rb[st][t][0].y, rb[st][t][1].x, rb[st][t][1].y);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]),
                                                                acc[mi][ni], 0, 0, 0);
    }
  }
};

template <typename T>
__global__ __launch_bounds__(NT, 2) void gemm_wgrad_kernel(const ZsGemmWgrad p, int ci_tiles, int rows_per_split,
                                                          int cout_r, int cin_r) {
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  constexpr int EPS = 16 / (int)sizeof(T);
  constexpr int SPR = 128 / EPS;               // 16-B segments per 128-element row
  constexpr int LPT = (WK * SPR) / NT;         // loads per thread per operand (2 bf16, 4 fp32)
  constexpr int PITCHW = WFrag<T>::PITCHW;
  constexpr int TILEW = WK * PITCHW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sY = smem;                    // [2][TILEW]
  unsigned char* sX = smem + 2 * TILEW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid, XCD-aware: the workgroups of one K split (same rows of dY and X, different tiles) get consecutive virtual
  // ids, i.e. the same XCD, so the operand rows are fetched into one L2 instead of all eight
  const int per_co = p.taps * ci_tiles;
  const int tiles = ((cout_r / 128) * per_co);
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vid / tiles, tile = vid - split * tiles;
  const int cot = tile / per_co;
  const int rem = tile - cot * per_co;
  const int tap = rem / ci_tiles, cit = rem - tap * ci_tiles;
  const int co0 = cot * 128, ci0 = cit * 128;
  const int M = p.B * p.T_out;
  const int nsplit = gridDim.x / tiles;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const T* __restrict__ dY = (const T*)p.dY;
  const T* __restrict__ X = (const T*)p.X;

  int lrow[LPT], lseg[LPT], lb[LPT], lt[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int idx = tid + NT * i;
    lrow[i] = idx / SPR; lseg[i] = idx - lrow[i] * SPR;
    const int m = mbeg + lrow[i];
    lb[i] = m / p.T_out; lt[i] = m - lb[i] * p.T_out;
  }
  uint4 vy[LPT], vx[LPT];
  // bias gradient: the (tap 0, ci-tile 0) workgroup of each co-tile also sums its dY tile over rows
  const bool do_bias = (p.db != nullptr) && tap == 0 && cit == 0;
  float bsum[LPT][EPS];
#pragma unroll
  for (int i = 0; i < LPT; ++i)
#pragma unroll
    for (int e = 0; e < EPS; ++e) bsum[i][e] = 0.f;
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int m = mb + lrow[i];
      const int cy = co0 + lseg[i] * EPS, cx = ci0 + lseg[i] * EPS;
      vy[i] = make_uint4(0, 0, 0, 0); vx[i] = make_uint4(0, 0, 0, 0);
      if (m < mend) {
        if (cy < p.y_cols) vy[i] = *reinterpret_cast<const uint4*>(dY + (int64_t)m * p.ldy + cy);
        bool ok; const int s = conv_src_row_g(0, p.pad_mode, p.stride, p.pad_left, p.T_in, lt[i], tap, ok, g2);
        if (ok && cx < p.x_cols)
          vx[i] = *reinterpret_cast<const uint4*>(X + (int64_t)lb[i] * p.x_batch_stride + (int64_t)s * p.ldx + cx);
      }
      lt[i] += WK;                               // next chunk: m += 32
      while (lt[i] >= p.T_out) { lt[i] -= p.T_out; ++lb[i]; }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      *reinterpret_cast<uint4*>(sY + buf * TILEW + lrow[i] * PITCHW + lseg[i] * 16) = vy[i];
      *reinterpret_cast<uint4*>(sX + buf * TILEW + lrow[i] * PITCHW + lseg[i] * 16) = vx[i];
      if (do_bias) {                                             // data has landed here anyway (wave-uniform branch)
        const uint32_t w4[4] = {vy[i].x, vy[i].y, vy[i].z, vy[i].w};
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[i][e] += __uint_as_float(w4[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bsum[i][2 * e] += __uint_as_float(w4[e] << 16);
            bsum[i][2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
          }
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (mend - mbeg + WK - 1) / WK;
  if (nchunks > 0) {
    gload(mbeg);
    swrite(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(mbeg + (c + 1) * WK);
    WFrag<T>::chunk(sY + buf * TILEW, sX + buf * TILEW, wm, wn, lane, acc);
    if (c + 1 < nchunks) swrite(buf ^ 1);
    __syncthreads();
  }

  if (do_bias) {
    // LDS is free now (the K loop ended with a barrier): red[32 rows][128 cols] fp32 = 16 KiB
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < LPT; ++i)
#pragma unroll
      for (int e = 0; e < EPS; ++e) red[lrow[i] * 128 + lseg[i] * EPS + e] = bsum[i][e];
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < WK; ++r) t += red[r * 128 + tid];
      float* bslab = (float*)p.workspace + (int64_t)nsplit * cout_r * p.taps * cin_r + (int64_t)split * cout_r;
      bslab[co0 + tid] = t;
    }
  }
  // slab[split][co][tap][ci], co < cout_r, ci < cin_r (full tiles: no guards needed).  Accumulators are restaged
  // through LDS so that each thread stores 8 consecutive ci (two 16-byte stores) of a co row.
  float* slab = (float*)p.workspace + (int64_t)split * cout_r * p.taps * cin_r;
  float* sC = reinterpret_cast<float*>(smem);
  __syncthreads();                                  // bias reduction above (if any) is done with LDS
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + ni * 32 + (lane & 31);
        sC[row * CPITCH + col] = acc[mi][ni][r];
      }
  __syncthreads();
  const int cg = tid & 15, rr0 = tid >> 4;
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const int row = rr0 + 16 * it;
    const float4 x0 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8);
    const float4 x1 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8 + 4);
    float* d = slab + ((int64_t)(co0 + row) * p.taps + tap) * cin_r + ci0 + cg * 8;
    *reinterpret_cast<float4*>(d) = x0;
    *reinterpret_cast<float4*>(d + 4) = x1;
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient on the quadrant ping-pong schedule of gemm_conv_p8_kernel (same regions, phases, DMA order and
// waits; read that comment first).  Tile = 256 co x 256 ci of one tap, K tile = 64 rows m.  Both operands are K(=row)-major
// in memory, so an LDS "row" is (64-channel block cb, row k): 128 B = 64 channels of dY (or gathered X) row k; the A region
// holds co blocks 0..3 (half-tiles A0 = blocks 0,1, A1 = 2,3), the B region ci blocks 0..3.  A 1-KiB DMA piece is 8 rows k
// of one channel block, so global reads stay 128 B contiguous per row, exactly as in the conv kernels.
// Fragments come from the transposing read ds_read_b64_tr_b16 (per 16-lane group: lane 4q+p addresses row q, channels
// 4p..4p+3 of a 4-row x 16-channel block; lane i receives channel i of the 4 rows): two reads per MFMA operand.  Rows are
// stored with the two 64-byte halves of a row swapped when (k>>1)&1 (done on the DMA source side), so the 4 rows x 64 B a
// 32-lane half touches fall on 4 distinct 64-byte bank windows.
// The (tap 0, ci tile 0) workgroup of each co tile also column-sums its dY tiles (bias gradient): 4 extra ds_read_b128
// per thread and K tile in the light phase 2, reduced through LDS at the end.
// ------------------------------------------------------------------------------------------------
#define ZS_DSRT(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define ZS_WP_READ_A(base)                                                                        \
  ZS_DSRT(ya00, sa0, base + 0);    ZS_DSRT(yb00, sa0, base + 512);  ZS_DSRT(ya01, sa1, base + 0);    ZS_DSRT(yb01, sa1, base + 512);  \
  ZS_DSRT(ya10, sa0, base + 2048); ZS_DSRT(yb10, sa0, base + 2560); ZS_DSRT(ya11, sa1, base + 2048); ZS_DSRT(yb11, sa1, base + 2560); \
  ZS_DSRT(ya20, sa0, base + 4096); ZS_DSRT(yb20, sa0, base + 4608); ZS_DSRT(ya21, sa1, base + 4096); ZS_DSRT(yb21, sa1, base + 4608); \
  ZS_DSRT(ya30, sa0, base + 6144); ZS_DSRT(yb30, sa0, base + 6656); ZS_DSRT(ya31, sa1, base + 6144); ZS_DSRT(yb31, sa1, base + 6656);
#define ZS_WP_READ_B(x, sbx)                                                                      \
  ZS_DSRT(x##a0, sbx, 0);    ZS_DSRT(x##b0, sbx, 512);  ZS_DSRT(x##a1, sbx, 2048); ZS_DSRT(x##b1, sbx, 2560); \
  ZS_DSRT(x##a2, sbx, 4096); ZS_DSRT(x##b2, sbx, 4608); ZS_DSRT(x##a3, sbx, 6144); ZS_DSRT(x##b3, sbx, 6656);
#define ZS_WP_MMA(alo, ahi, blo, bhi, c)                                                          \
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(alo.x, alo.y, ahi.x, ahi.y)), \
                                               __builtin_bit_cast(bf16x8, make_uint4(blo.x, blo.y, bhi.x, bhi.y)), c, 0, 0, 0)
#define ZS_WP_DMA(cond, src, dst)                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if (cond) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, 0, 0);             \
  __builtin_amdgcn_sched_barrier(0);
#define ZS_WP_QUAD(c0, c1, x, cond, src0, dst0, src1, dst1)                                        \
  __builtin_amdgcn_s_setprio(1);                                                                  \
  ZS_WP_MMA(ya00, yb00, x##a0, x##b0, c0); ZS_WP_MMA(ya01, yb01, x##a0, x##b0, c1);               \
  ZS_WP_DMA(cond, src0, dst0)                                                                     \
  ZS_WP_MMA(ya10, yb10, x##a1, x##b1, c0); ZS_WP_MMA(ya11, yb11, x##a1, x##b1, c1); ZS_WP_MMA(ya20, yb20, x##a2, x##b2, c0); \
  ZS_WP_DMA(cond, src1, dst1)                                                                     \
  ZS_WP_MMA(ya21, yb21, x##a2, x##b2, c1); ZS_WP_MMA(ya30, yb30, x##a3, x##b3, c0); ZS_WP_MMA(ya31, yb31, x##a3, x##b3, c1); \
  __builtin_amdgcn_s_setprio(0);

__global__ __launch_bounds__(PNT, 2) void gemm_wgrad_p8_kernel(const ZsGemmWgrad p, int ci_tiles, int rows_per_split,
                                                              int cout_r, int cin_r) {
  const Geom2 g2 = make_geom2(p.w_in, p.w_out, p.taps_h, p.T_in);
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // 1-D grid, XCD-aware: all tiles of one K split (same rows of dY and X) land on one XCD (see gemm_wgrad_kernel)
  const int per_co = p.taps * ci_tiles;
  const int tiles = (cout_r / 256) * per_co;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vid / tiles, tile = vid - split * tiles;
  const int cot = tile / per_co;
  const int rem = tile - cot * per_co;
  const int tap = rem / ci_tiles, cit = rem - tap * ci_tiles;
  const int co0 = cot * 256, ci0 = cit * 256;
  const int M = p.B * p.T_out;
  const int nsplit = gridDim.x / tiles;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int nk = (mend - mbeg + 63) / 64;
  const T* __restrict__ dY = (const T*)p.dY;
  const T* __restrict__ X = (const T*)p.X;
  const T* zline = reinterpret_cast<const T*>(zs_zero_line);

  // loader: piece j = 0..3 covers LDS rows (j>>1)*128 + wave*16 + (j&1)*8 + lane/8 = (channel block (j>>1)*2 + wave/4, row k)
  const int lrow = lane >> 3, slot = lane & 7;
  const int k0 = (wave & 3) * 16 + lrow, k1 = k0 + 8;                       // the two rows k this lane loads (pieces 0,2 / 1,3)
  const int cbl = wave >> 2;                                                // channel block of pieces 0,1; pieces 2,3: +2
  const int sg0 = slot ^ (((k0 >> 1) & 1) << 2), sg1 = slot ^ (((k1 >> 1) & 1) << 2);   // source 16-B segment of the 128-B row
  // column offsets (elements) and validity of the four pieces, per operand
  const int yc0 = co0 + cbl * 64 + sg0 * 8, yc1 = co0 + cbl * 64 + sg1 * 8;   // pieces 0,1 ; pieces 2,3: +128
  const int xc0 = ci0 + cbl * 64 + sg0 * 8, xc1 = ci0 + cbl * 64 + sg1 * 8;
  auto y_ptr = [&](int m, int col) -> const T* {
    return (m < mend && col < p.y_cols) ? dY + (int64_t)m * p.ldy + col : zline;
  };
  auto x_row = [&](int m, bool& ok) -> const T* {                            // gathered source row of output row m for this tap
    ok = false;
    if (m >= mend) return zline;
    const int b = m / p.T_out, t = m - b * p.T_out;
    const int srow = conv_src_row_g(0, p.pad_mode, p.stride, p.pad_left, p.T_in, t, tap, ok, g2);
    return X + (int64_t)b * p.x_batch_stride + (int64_t)srow * p.ldx;
  };

  unsigned char* const ldsA = smem + wave * 2048;
  unsigned char* const ldsB = smem + 32768 + wave * 2048;

  f32x16 c000, c001, c010, c011, c100, c101, c110, c111;     // c[mq][nq][mi]: co = wr*128 + mq*64 + mi*32.., ci = wc*64 + nq*32..
#pragma unroll
  for (int r = 0; r < 16; ++r) { c000[r] = 0.f; c001[r] = 0.f; c010[r] = 0.f; c011[r] = 0.f; c100[r] = 0.f; c101[r] = 0.f; c110[r] = 0.f; c111[r] = 0.f; }

  // fragment addresses.  16-lane group gi: channels 16*(gi&1).., rows 8*(gi>>1)..; lane 4q+pp in it: row q, channels 4pp..
  const int gi = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = ((q >> 1) & 1) << 2;                                         // (k>>1)&1 of every row this lane addresses
  const int segl = 2 * (gi & 1) + (pp >> 1);                                  // 16-B segment within a 32-channel MFMA tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned rowoff = (unsigned)((8 * (gi >> 1) + q) * 128 + (pp & 1) * 8);
  const unsigned a_l0 = lds0 + (unsigned)(wr * 2 * 64 * 128) + rowoff + (unsigned)(((0 + segl) ^ sw) * 16);   // mi = 0
  const unsigned a_l1 = lds0 + (unsigned)(wr * 2 * 64 * 128) + rowoff + (unsigned)(((4 + segl) ^ sw) * 16);   // mi = 1
  const unsigned b_l0 = lds0 + 32768u + (unsigned)(wc * 64 * 128) + rowoff + (unsigned)(((0 + segl) ^ sw) * 16);   // nq = 0
  const unsigned b_l1 = lds0 + 32768u + (unsigned)(wc * 64 * 128) + rowoff + (unsigned)(((4 + segl) ^ sw) * 16);   // nq = 1
  const int grp1 = wr;

  // bias gradient (wave-uniform): group wr sums the A half it reads; thread tg: 8 channels (seg16), rows (tg>>4) + 16 i
  const bool do_bias = (p.db != nullptr) && tap == 0 && cit == 0;
  const int tg = tid & 255, seg16 = tg & 15, kr0 = tg >> 4;
  const unsigned bias_addr = lds0 + (unsigned)(((wr * 2 + (seg16 >> 3)) * 64 + kr0) * 128 + (((seg16 & 7) ^ (((kr0 >> 1) & 1) << 2)) * 16));
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  // prologue: A(0) B(0) -> stage 0, B(1) -> stage 1
  {
    __builtin_amdgcn_global_load_lds((gptr_t)y_ptr(mbeg + k0, yc0), (lptr_t)(ldsA), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)y_ptr(mbeg + k1, yc1), (lptr_t)(ldsA + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)y_ptr(mbeg + k0, yc0 + 128), (lptr_t)(ldsA + 16384), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)y_ptr(mbeg + k1, yc1 + 128), (lptr_t)(ldsA + 16384 + 1024), 16, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tt < nk) {
        bool ok0, ok1;
        const T* r0 = x_row(mbeg + tt * 64 + k0, ok0);
        const T* r1 = x_row(mbeg + tt * 64 + k1, ok1);
        unsigned char* d = ldsB + tt * PSTAGE;
        __builtin_amdgcn_global_load_lds((gptr_t)((ok0 && xc0 < p.x_cols) ? r0 + xc0 : zline), (lptr_t)(d), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)((ok1 && xc1 < p.x_cols) ? r1 + xc1 : zline), (lptr_t)(d + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)((ok0 && xc0 + 128 < p.x_cols) ? r0 + xc0 + 128 : zline), (lptr_t)(d + 16384), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)((ok1 && xc1 + 128 < p.x_cols) ? r1 + xc1 + 128 : zline), (lptr_t)(d + 16384 + 1024), 16, 0, 0);
      }
    }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp1) __builtin_amdgcn_s_barrier();

  uint2 ya00, yb00, ya01, yb01, ya10, yb10, ya11, yb11, ya20, yb20, ya21, yb21, ya30, yb30, ya31, yb31;   // y{a,b}[ks][mi]: rows +0..3 / +4..7
  uint2 f0a0, f0b0, f0a1, f0b1, f0a2, f0b2, f0a3, f0b3;                                                   // B nq=0: [ks]
  uint2 f1a0, f1b0, f1a1, f1b1, f1a2, f1b2, f1a3, f1b3;                                                   // B nq=1
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    const unsigned so = (unsigned)(st * PSTAGE);
    const unsigned sa0 = a_l0 + so, sa1 = a_l1 + so, sb0 = b_l0 + so, sb1 = b_l1 + so;
    const bool has_a = kt + 1 < nk, has_b = kt + 2 < nk;
    unsigned char* const dA = ldsA + (st ^ 1) * PSTAGE;
    unsigned char* const dB = ldsB + st * PSTAGE;
    const int ma = mbeg + (kt + 1) * 64, mb = mbeg + (kt + 2) * 64;
    // ---- phase 1: quadrant (0,0); A0(kt+1)
    ZS_WP_READ_A(0)
    ZS_WP_READ_B(f0, sb0)
    const T* s0 = y_ptr(ma + k0, yc0);
    const T* s1 = y_ptr(ma + k1, yc1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_WP_QUAD(c000, c001, f0, has_a, s0, dA, s1, dA + 1024)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: quadrant (0,1); A1(kt+1); bias partial sums of this tile
    ZS_WP_READ_B(f1, sb1)
    s0 = y_ptr(ma + k0, yc0 + 128);
    s1 = y_ptr(ma + k1, yc1 + 128);
    if (do_bias) {
      u32x4_t v0, v1, v2, v3;
      const unsigned ba = bias_addr + so;
      ZS_DSR(v0, ba, 0); ZS_DSR(v1, ba, 2048); ZS_DSR(v2, ba, 4096); ZS_DSR(v3, ba, 6144);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        bsum[2 * (i & 3)] += __uint_as_float(w[i] << 16);
        bsum[2 * (i & 3) + 1] += __uint_as_float(w[i] & 0xffff0000u);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_WP_QUAD(c010, c011, f1, has_a, s0, dA + 16384, s1, dA + 16384 + 1024)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: quadrant (1,1); B0(kt+2)
    ZS_WP_READ_A(8192)
    bool ok0, ok1;
    const T* r0 = x_row(mb + k0, ok0);
    const T* r1 = x_row(mb + k1, ok1);
    s0 = (ok0 && xc0 < p.x_cols) ? r0 + xc0 : zline;
    s1 = (ok1 && xc1 < p.x_cols) ? r1 + xc1 : zline;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_WP_QUAD(c110, c111, f1, has_b, s0, dB, s1, dB + 1024)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 4: quadrant (1,0); B1(kt+2); the counted waits for tile kt+1
    s0 = (ok0 && xc0 + 128 < p.x_cols) ? r0 + xc0 + 128 : zline;
    s1 = (ok1 && xc1 + 128 < p.x_cols) ? r1 + xc1 + 128 : zline;
    if (grp1) {
      if (has_b) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ZS_WP_QUAD(c100, c101, f0, has_b, s0, dB + 16384, s1, dB + 16384 + 1024)
    if (!grp1) {
      if (has_b) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!grp1) __builtin_amdgcn_s_barrier();          // pairs with group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (do_bias) {
    // red[group][16 row lanes][128 channels] fp32 = 16 KiB; fixed summation order -> deterministic
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(wr * 16 + kr0) * 128 + seg16 * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[((tid >> 7) * 16 + r) * 128 + (tid & 127)];
      float* bslab = (float*)p.workspace + (int64_t)nsplit * cout_r * p.taps * cin_r + (int64_t)split * cout_r;
      bslab[co0 + tid] = t;
    }
    __syncthreads();
  }
  // slab[split][co][tap][ci] (co < cout_r, ci < cin_r: whole tiles), staged through LDS in two 128-ci halves
  float* slab = (float*)p.workspace + (int64_t)split * cout_r * p.taps * cin_r;
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if ((wc >> 1) == h) {
      const int cb = (wc & 1) * 64 + (lane & 31);
      const int rbase = wr * 128 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        sC[(rbase + ro) * CPITCH + cb] = c000[r];            sC[(rbase + 32 + ro) * CPITCH + cb] = c001[r];
        sC[(rbase + ro) * CPITCH + cb + 32] = c010[r];       sC[(rbase + 32 + ro) * CPITCH + cb + 32] = c011[r];
        sC[(rbase + 64 + ro) * CPITCH + cb] = c100[r];       sC[(rbase + 96 + ro) * CPITCH + cb] = c101[r];
        sC[(rbase + 64 + ro) * CPITCH + cb + 32] = c110[r];  sC[(rbase + 96 + ro) * CPITCH + cb + 32] = c111[r];
      }
    }
    __syncthreads();
    const int cg = tid & 15, rr0 = tid >> 4;          // 16 groups of 8 ci x 32 co rows per pass
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
      const int row = rr0 + 32 * it;
      const float4 x0 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(sC + row * CPITCH + cg * 8 + 4);
      float* d = slab + ((int64_t)(co0 + row) * p.taps + tap) * cin_r + ci0 + h * 128 + cg * 8;
      *reinterpret_cast<float4*>(d) = x0;
      *reinterpret_cast<float4*>(d + 4) = x1;
    }
    __syncthreads();
  }
}

// Fixed-order sum of the split-K slabs (deterministic): one thread per 4 consecutive ci of a (co, tap) row, eight slab
// loads in flight per thread (the slab pitch cin_r is a multiple of 128, so the float4 loads are aligned and in bounds).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ZsGemmWgrad p, int splits, int cout_r, int cin_r) {
  const int cin4 = (p.Cin + 3) >> 2;
  const int64_t total = (int64_t)p.Cout * p.taps * cin4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % cin4);
  const int64_t rowi = idx / cin4;
  const int j = (int)(rowi % p.taps);
  const int co = (int)(rowi / p.taps);
  const int half = p.Cout >> 1;
  const int cop = p.co_split2 ? ((co & 1) * half + (co >> 1)) : co;
  const int64_t sstride = (int64_t)cout_r * p.taps * cin_r;
  const float* ws = (const float*)p.workspace + ((int64_t)cop * p.taps + j) * cin_r + c4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(ws + (int64_t)(k + u) * sstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)k * sstride);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float r[4] = {acc.x, acc.y, acc.z, acc.w};
  float* d = p.dW + (int64_t)co * p.so + (int64_t)j * p.sj;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ci = c4 * 4 + e;
    if (ci < p.Cin) {
      float* q = d + (int64_t)ci * p.si;
      *q = p.accumulate ? (*q + r[e]) : r[e];
    }
  }
  if (p.db != nullptr && c4 == 0 && j == 0) {
    const float* bs = (const float*)p.workspace + (int64_t)splits * sstride + cop;
    float t = 0.f;
    for (int q2 = 0; q2 < splits; ++q2) t += bs[(int64_t)q2 * cout_r];
    p.db[co] = p.accumulate ? (p.db[co] + t) : t;
  }
}

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// Kernel-selection knobs: process-wide relaxed atomics (settable from any thread; every launch reads each knob once into a
// local).  Defaults come from the environment on first use.
struct Knob {
  std::atomic<int> v{-1};
  const char* env; int dflt;
  Knob(const char* e, int d) : env(e), dflt(d) {}
  int get() {
    int x = v.load(std::memory_order_relaxed);
    if (x < 0) { x = env_int(env, dflt); if (x < 0) x = 0; v.store(x, std::memory_order_relaxed); }
    return x;
  }
  int set(int nv) { const int old = get(); v.store(nv < 0 ? 0 : nv, std::memory_order_relaxed); return old; }
};
Knob g_dma64("ZS_GEMM_DMA64", 1);                 // 64-row tiles of the LDS-DMA kernel for under-filled grids
Knob g_dma64_max_tiles("ZS_GEMM_DMA64_MAX_TILES", 256);   // ... when twice the 128-row tile count is at most this
Knob g_wgrad_p8("ZS_WGRAD_P8", 1);
Knob g_wgrad_wgs("ZS_WGRAD_WGS", 256);      // workgroups one weight-gradient launch aims for (split-K plan of the 256x256 kernel)
// workgroups a 128x128 weight-gradient launch aims for.  768 (three per CU-pair slot) is best for a launch alone; the launches of
// this kernel run four side by side (the conv bank's seven at the end of the step), where fewer, deeper splits write and
// re-read less slab: 448 measured 10.93 against 10.98 ms/step (384-512 equal, 256 and 1024 worse)
Knob g_wgrad_wgs128("ZS_WGRAD_WGS128", 448);
Knob g_wgrad_slab_cost("ZS_WGRAD_SLAB_COST", 13);   // cost of one split's slab write + re-read in K-tile times
Knob g_wgrad_waste("ZS_WGRAD_P8_WASTE", 135);       // 256x256 weight-gradient tiles only while padded/real output area <= this / 100

struct WgradPlan { int p8, splits, tile, co_tiles, ci_tiles, cout_r, cin_r, rows_per_split; };

// Split-K plan.  128x128 tiles (fp32, small outputs): ~448 workgroups (wgrad_wgs128).  256x256 ping-pong tiles (bf16): one workgroup per
// CU and round, each split pays a 256-KiB slab write + re-read (~13 K-tile times at 1/256 of HBM), so minimise
// rounds x (K tiles per split + 13).
WgradPlan wgrad_plan(const ZsGemmWgrad* p) {
  WgradPlan w;
  const int wgrad_p8 = g_wgrad_p8.get();
  const int64_t M = (int64_t)p->B * p->T_out;
  const int64_t t256 = (int64_t)((p->Cout + 255) / 256) * p->taps * ((p->Cin + 255) / 256);
  // padding waste of 256-wide tiles must stay small, and there must be enough K per workgroup to amortise the slab
  const double waste = (double)(((p->Cout + 255) / 256) * 256) * (((p->Cin + 255) / 256) * 256) / ((double)p->Cout * p->Cin);
  w.p8 = wgrad_p8 && p->dtype == ZS_BF16 && waste <= 0.01 * (double)g_wgrad_waste.get() && M >= 2048 && (wgrad_p8 > 1 || t256 * M >= (int64_t)16 * 8192);
  if (wgrad_p8 > 1 && p->dtype == ZS_BF16) w.p8 = 1;                   // forced (tests)
  w.tile = w.p8 ? 256 : 128;
  w.co_tiles = (p->Cout + w.tile - 1) / w.tile; w.ci_tiles = (p->Cin + w.tile - 1) / w.tile;
  w.cout_r = w.co_tiles * w.tile; w.cin_r = w.ci_tiles * w.tile;
  const int64_t tiles = (int64_t)w.co_tiles * p->taps * w.ci_tiles;
  if (p->splits > 0) {
    w.splits = p->splits;
  } else if (w.p8) {
    const int64_t kt = (M + 63) / 64;
    const int64_t wgs = g_wgrad_wgs.get() > 0 ? g_wgrad_wgs.get() : 256;
    const double slab_cost = (double)g_wgrad_slab_cost.get();
    double best = 1e30; int bs = 1;
    for (int sp = 1; sp <= 64 && (int64_t)sp * 4 <= kt; ++sp) {
      const double rounds = (double)((tiles * sp + wgs - 1) / wgs);
      const double cost = rounds * ((double)((kt + sp - 1) / sp) + slab_cost);
      if (cost < best - 1e-9) { best = cost; bs = sp; }
    }
    w.splits = bs;
  } else {
    const int64_t target = g_wgrad_wgs128.get() > 0 ? g_wgrad_wgs128.get() : 448;
    int64_t sp = (target + tiles - 1) / tiles;
    int64_t maxs = (M + 255) / 256;
    if (sp > maxs) sp = maxs;
    if (sp > 1024) sp = 1024;        // (a one-tile output over millions of rows -- the first stage-2 layer: 64 x 25 over 2.1 M rows --
    if (sp < 1) sp = 1;              //  needs hundreds of splits to fill the chip; the cap used to be 64: 64 workgroups, 1.2 ms)
    w.splits = (int)sp;
  }
  const int gran = w.p8 ? 64 : WK;
  int rps = (int)((M + w.splits - 1) / w.splits);
  w.rows_per_split = ((rps + gran - 1) / gran) * gran;
  return w;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

Knob g_use_dma("ZS_GEMM_DMA", 1), g_use_ring("ZS_GEMM_RING", 1), g_ring_min_tiles("ZS_GEMM_RING_MIN_TILES", 256), g_use_pp("ZS_GEMM_PP", 1),
    g_use_p8("ZS_GEMM_P8", 1), g_p8_min_tiles("ZS_GEMM_P8_MIN_TILES", 200);

// one-time raise of the dynamic-LDS limit of the big-tile kernels (std::call_once: launches may come from several threads)
std::once_flag g_lds_attr_once;
void set_lds_attrs() {
  const void* big[] = {reinterpret_cast<const void*>(gemm_conv_p8m16_kernel<float>), reinterpret_cast<const void*>(gemm_conv_p8m16_kernel<bf16_t>),
                       reinterpret_cast<const void*>(gemm_wgrad_p8_kernel)};
  for (const void* f : big) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
  const void* ring[] = {reinterpret_cast<const void*>(gemm_conv_ring_kernel<float, 0>), reinterpret_cast<const void*>(gemm_conv_ring_kernel<bf16_t, 0>),
                        reinterpret_cast<const void*>(gemm_conv_ring_kernel<float, 1>), reinterpret_cast<const void*>(gemm_conv_ring_kernel<bf16_t, 1>)};
  for (const void* f : ring) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------
extern "C" int zs_gemm_conv(const ZsGemmConv* p, void* stream) {
  ZS_REQUIRE(p && p->A && p->W && (p->out || p->out2), "zs_gemm_conv: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_gemm_conv: bad dtype %d", p->dtype);
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  const int kc = 128 / es;
  ZS_REQUIRE(p->B > 0 && p->T_in > 0 && p->T_out > 0 && p->N > 0 && p->taps > 0 && p->stride > 0, "zs_gemm_conv: bad sizes");
  ZS_REQUIRE(p->cin_pad > 0 && p->cin_pad % kc == 0, "zs_gemm_conv: cin_pad %d must be a multiple of %d (one 128-byte chunk)", p->cin_pad, kc);
  ZS_REQUIRE(p->ldw >= (int64_t)p->taps * p->cin_pad && (p->ldw * es) % 16 == 0, "zs_gemm_conv: ldw %lld < taps*cin_pad",
             (long long)p->ldw);
  ZS_REQUIRE(p->n_pad % 128 == 0 && p->n_pad >= p->N, "zs_gemm_conv: n_pad %d must be a multiple of 128 >= N %d", p->n_pad, p->N);
  ZS_REQUIRE(aligned16(p->A) && aligned16(p->W) && (p->lda * es) % 16 == 0 && (p->a_batch_stride * es) % 16 == 0 &&
                 (p->a_gstride * es) % 16 == 0 && (p->w_gstride * es) % 16 == 0,
             "zs_gemm_conv: operands must be 16-byte aligned (A=%p lda=%lld)", p->A, (long long)p->lda);
  if (p->w_in > 0) {
    ZS_REQUIRE((int64_t)p->T_out * p->w_out < (1ll << 32), "zs_gemm_conv: 2-D image too large (T_out %d x w_out %d)", p->T_out, p->w_out);
    ZS_REQUIRE(p->w_out > 0 && p->taps_h > 0 && p->taps % p->taps_h == 0 && p->T_in % p->w_in == 0 && p->T_out % p->w_out == 0 && !p->lengths,
               "zs_gemm_conv: 2-D geometry (w_in %d, w_out %d, taps_h %d, taps %d, T_in %d, T_out %d)", p->w_in, p->w_out, p->taps_h, p->taps,
               p->T_in, p->T_out);
    if (p->gather == 0 && p->pad_mode == ZS_PAD_REFLECT) {
      const int h_in = p->T_in / p->w_in, h_out = p->T_out / p->w_out, taps_w = p->taps / p->taps_h;
      const int pr_h = (h_out - 1) * p->stride + p->taps_h - 1 - p->pad_left - (h_in - 1);
      const int pr_w = (p->w_out - 1) * p->stride + taps_w - 1 - p->pad_left - (p->w_in - 1);
      ZS_REQUIRE(p->pad_left < h_in && pr_h < h_in && p->pad_left < p->w_in && pr_w < p->w_in,
                 "zs_gemm_conv: Padding size should be less than the corresponding input dimension (2-D: pad %d, H %d, W %d)", p->pad_left, h_in, p->w_in);
    }
  } else if (p->gather == 0 && p->pad_mode == ZS_PAD_REFLECT) {
    const int pad_r = (p->T_out - 1) * p->stride + p->taps - 1 - p->pad_left - (p->T_in - 1);
    ZS_REQUIRE(p->pad_left < p->T_in && pad_r < p->T_in,
               "zs_gemm_conv: Padding size should be less than the corresponding input dimension (pad %d/%d, T %d)",
               p->pad_left, pad_r, p->T_in);
  }
  if (p->store_mode == ZS_STORE_SPLIT2 || (p->out2 && p->store_mode2 == ZS_STORE_SPLIT2))
    ZS_REQUIRE(p->N % 2 == 0, "zs_gemm_conv: SPLIT2 needs even N");
  ZS_REQUIRE(!(p->pre_vec || p->vec2) || p->vec_idx, "zs_gemm_conv: vec_idx missing");
  ZS_REQUIRE(!p->colsum || (128 % p->T_out == 0 && p->colsum_col0 >= 0 && p->colsum_col0 < p->N && !p->bias && !p->pre_vec &&
                            p->act == ZS_ACT_NONE && p->groups <= 1),
             "zs_gemm_conv: colsum needs T_out (%d) to divide 128, 0 <= colsum_col0 < N and no bias / pre_vec / activation / groups", p->T_out);
  ZS_REQUIRE(!p->lengths || (p->gather == 0 && p->B < 65536 && p->T_in < 32768 && p->pad_right >= 0 && p->colsum == nullptr),
             "zs_gemm_conv: lengths (ragged batch) needs gather 0, B < 65536, T_in < 32768, pad_right >= 0 and no colsum");
  const int groups = p->groups > 0 ? p->groups : 1;
  const int64_t M = (int64_t)p->B * p->T_out;
  const int64_t tiles = ((M + BM - 1) / BM) * ((p->N + BN - 1) / BN);
  ZS_REQUIRE(tiles < (1ll << 31), "zs_gemm_conv: grid too large");
  dim3 grid((unsigned)tiles, 1, (unsigned)groups);
  hipStream_t s = (hipStream_t)stream;
  std::call_once(g_lds_attr_once, set_lds_attrs);
  const int use_dma = g_use_dma.get(), use_ring = g_use_ring.get(), use_p8 = g_use_p8.get(), use_pp = g_use_pp.get();
  const int64_t ring_tiles = ((M + RBM - 1) / RBM) * ((p->N + BN - 1) / BN);
  const int64_t p8_tiles = ((M + PBM - 1) / PBM) * ((p->N + PBN - 1) / PBN);
  if (use_dma && use_p8 && p->n_pad % PBN == 0 && p8_tiles >= g_p8_min_tiles.get() && p8_tiles < (1ll << 31)) {
    // enough 256x256 tiles for most of the chip: quadrant ping-pong kernel
    dim3 pgrid((unsigned)p8_tiles, 1, (unsigned)groups);
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(gemm_conv_p8m16_kernel<float>, pgrid, dim3(PNT), P8_LDS, s, *p);
    else hipLaunchKernelGGL(gemm_conv_p8m16_kernel<bf16_t>, pgrid, dim3(PNT), P8_LDS, s, *p);
  } else if (use_dma && use_ring && ring_tiles >= g_ring_min_tiles.get() && ring_tiles < (1ll << 31)) {
    // enough 256x128 tiles to give every CU one workgroup: 3-stage ring kernel
    dim3 rgrid((unsigned)ring_tiles, 1, (unsigned)groups);
    if (use_pp) {
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((gemm_conv_ring_kernel<float, 1>), rgrid, dim3(RNT), RING_LDS, s, *p);
      else hipLaunchKernelGGL((gemm_conv_ring_kernel<bf16_t, 1>), rgrid, dim3(RNT), RING_LDS, s, *p);
    } else {
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((gemm_conv_ring_kernel<float, 0>), rgrid, dim3(RNT), RING_LDS, s, *p);
      else hipLaunchKernelGGL((gemm_conv_ring_kernel<bf16_t, 0>), rgrid, dim3(RNT), RING_LDS, s, *p);
    }
  } else if (use_dma) {
    // 64-row tiles where the 128-row grid would leave half the chip without a workgroup
    const int64_t tiles64 = ((M + 63) / 64) * ((p->N + BN - 1) / BN);
    if (g_dma64.get() && tiles * 2 <= g_dma64_max_tiles.get() && tiles64 < (1ll << 31) && (p->colsum == nullptr || 64 % p->T_out == 0)) {
      dim3 g64((unsigned)tiles64, 1, (unsigned)groups);
      const size_t lds = (2 * 64 * 128 + 2 * DTILE > EPI_LDS_BYTES / 2) ? 2 * 64 * 128 + 2 * DTILE : EPI_LDS_BYTES / 2;
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((gemm_conv_dma_kernel<float, 64>), g64, dim3(NT), lds, s, *p);
      else hipLaunchKernelGGL((gemm_conv_dma_kernel<bf16_t, 64>), g64, dim3(NT), lds, s, *p);
      return zs_check_launch("zs_gemm_conv");
    }
    const size_t lds = (4 * DTILE > EPI_LDS_BYTES) ? 4 * DTILE : EPI_LDS_BYTES;
    if (p->dtype == ZS_F32) hipLaunchKernelGGL((gemm_conv_dma_kernel<float, 128>), grid, dim3(NT), lds, s, *p);
    else hipLaunchKernelGGL((gemm_conv_dma_kernel<bf16_t, 128>), grid, dim3(NT), lds, s, *p);
  } else {
    const size_t lds = 4 * TILE_BYTES;
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(gemm_conv_kernel<float>, grid, dim3(NT), lds, s, *p);
    else hipLaunchKernelGGL(gemm_conv_kernel<bf16_t>, grid, dim3(NT), lds, s, *p);
  }
  return zs_check_launch("zs_gemm_conv");
}

extern "C" int zs_set_option(const char* key, int value) {
  Knob* slot = nullptr;
  if (key && !strcmp(key, "gemm_dma")) slot = &g_use_dma;
  else if (key && !strcmp(key, "gemm_ring")) slot = &g_use_ring;
  else if (key && !strcmp(key, "gemm_ring_min_tiles")) slot = &g_ring_min_tiles;
  else if (key && !strcmp(key, "gemm_pp")) slot = &g_use_pp;
  else if (key && !strcmp(key, "gemm_p8")) slot = &g_use_p8;
  else if (key && !strcmp(key, "gemm_p8_min_tiles")) slot = &g_p8_min_tiles;
  else if (key && !strcmp(key, "gemm_dma64")) slot = &g_dma64;
  else if (key && !strcmp(key, "gemm_dma64_max_tiles")) slot = &g_dma64_max_tiles;
  else if (key && !strcmp(key, "wgrad_p8")) slot = &g_wgrad_p8;
  else if (key && !strcmp(key, "wgrad_wgs")) slot = &g_wgrad_wgs;
  else if (key && !strcmp(key, "wgrad_wgs128")) slot = &g_wgrad_wgs128;
  else if (key && !strcmp(key, "wgrad_slab_cost")) slot = &g_wgrad_slab_cost;
  else if (key && !strcmp(key, "wgrad_p8_waste")) slot = &g_wgrad_waste;
  if (key && !strcmp(key, "gru_persist")) return zs_gru_persist_option(value);
  if (key && !strcmp(key, "gru_wide")) return zs_gru_wide_option(value);
  if (key && !strcmp(key, "gru_spin_limit")) return zs_gru_spin_limit_option(value);
  if (key && !strcmp(key, "gl_prefetch")) return zs_gl_prefetch_option(value);
  if (key && !strcmp(key, "gl_chains")) return zs_gl_chains_option(value);
  if (key && !strcmp(key, "norm_wide")) return zs_norm_wide_option(value);
  if (key && !strncmp(key, "norm_lim", 8) && key[8] >= '0' && key[8] <= '2' && !key[9]) return zs_norm_lim_option(key[8] - '0', value);
  if (!slot) { zs_set_error("zs_set_option: unknown key %s", key ? key : "(null)"); return ZS_EINVAL; }
  return slot->set(value);
}

extern "C" size_t zs_gemm_wgrad_workspace_bytes(const ZsGemmWgrad* p) {
  if (!p) return 0;
  const WgradPlan w = wgrad_plan(p);
  return (size_t)w.splits * ((size_t)w.cout_r * p->taps * w.cin_r + w.cout_r) * sizeof(float);
}

extern "C" int zs_gemm_wgrad(const ZsGemmWgrad* p, void* stream) {
  ZS_REQUIRE(p && p->dY && p->X && p->dW && p->workspace, "zs_gemm_wgrad: null operand");
  ZS_REQUIRE(aligned16(p->workspace), "zs_gemm_wgrad: workspace must be 16-byte aligned");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_gemm_wgrad: bad dtype");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE(p->B > 0 && p->T_in > 0 && p->T_out > 0 && p->Cout > 0 && p->Cin > 0 && p->taps > 0 && p->stride > 0,
             "zs_gemm_wgrad: bad sizes");
  ZS_REQUIRE(aligned16(p->dY) && aligned16(p->X) && (p->ldy * es) % 16 == 0 && (p->ldx * es) % 16 == 0 &&
                 (p->x_batch_stride * es) % 16 == 0 && (p->y_cols * es) % 16 == 0 && (p->x_cols * es) % 16 == 0,
             "zs_gemm_wgrad: operands must be 16-byte aligned");
  ZS_REQUIRE(p->y_cols <= p->ldy && p->x_cols <= p->ldx && p->Cout <= p->y_cols && p->Cin <= p->x_cols, "zs_gemm_wgrad: column bounds");
  ZS_REQUIRE(!p->co_split2 || p->Cout % 2 == 0, "zs_gemm_wgrad: SPLIT2 needs even Cout");
  ZS_REQUIRE(p->w_in == 0 || (p->w_in > 0 && p->w_out > 0 && p->taps_h > 0 && p->taps % p->taps_h == 0 && p->T_in % p->w_in == 0 &&
                              p->T_out % p->w_out == 0 && (int64_t)p->T_out * p->w_out < (1ll << 32)),
             "zs_gemm_wgrad: 2-D geometry (w_in %d, w_out %d, taps_h %d, taps %d, T_in %d, T_out %d)", p->w_in, p->w_out, p->taps_h, p->taps,
             p->T_in, p->T_out);
  const WgradPlan w = wgrad_plan(p);
  const int splits = w.splits, co_tiles = w.co_tiles, ci_tiles = w.ci_tiles, cout_r = w.cout_r, cin_r = w.cin_r;
  const int rows_per_split = w.rows_per_split;
  const size_t need = (size_t)splits * ((size_t)cout_r * p->taps * cin_r + cout_r) * sizeof(float);
  if (p->workspace_bytes < need) {
    zs_set_error("zs_gemm_wgrad: workspace %zu < %zu bytes", p->workspace_bytes, need);
    return ZS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  ZS_REQUIRE((int64_t)co_tiles * p->taps * ci_tiles * splits < (1ll << 31), "zs_gemm_wgrad: grid too large");
  dim3 grid((unsigned)(co_tiles * p->taps * ci_tiles * splits), 1, 1);
  std::call_once(g_lds_attr_once, set_lds_attrs);
  if (w.p8) {
    hipLaunchKernelGGL(gemm_wgrad_p8_kernel, grid, dim3(PNT), P8_LDS, s, *p, ci_tiles, rows_per_split, cout_r, cin_r);
  } else if (p->dtype == ZS_F32) {
    const size_t lds = (4 * WK * WFrag<float>::PITCHW > EPI_LDS_BYTES) ? 4 * WK * WFrag<float>::PITCHW : EPI_LDS_BYTES;
    hipLaunchKernelGGL(gemm_wgrad_kernel<float>, grid, dim3(NT), lds, s, *p, ci_tiles, rows_per_split, cout_r, cin_r);
  } else {
    const size_t lds = (4 * WK * WFrag<bf16_t>::PITCHW > EPI_LDS_BYTES) ? 4 * WK * WFrag<bf16_t>::PITCHW : EPI_LDS_BYTES;
    hipLaunchKernelGGL(gemm_wgrad_kernel<bf16_t>, grid, dim3(NT), lds, s, *p, ci_tiles, rows_per_split, cout_r, cin_r);
  }
  int rc = zs_check_launch("zs_gemm_wgrad");
  if (rc) return rc;
  const int64_t relems = (int64_t)p->Cout * p->taps * ((p->Cin + 3) / 4);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((relems + 255) / 256)), dim3(256), 0, s, *p, splits, cout_r, cin_r);
  return zs_check_launch("zs_gemm_wgrad.reduce");
}
