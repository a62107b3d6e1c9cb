// zs_elem.hip -- HBM-bound elementwise / reduction kernels of the autoencoder path: weight packing,
// casts, speaker-embedding adds and gradient scatter, the MBV discretiser, L1 loss, grad-norm, Adam, CE.
#include <string.h>

#include "zs_common.h"

namespace {

constexpr int NTE = 256;

template <typename T> __device__ __forceinline__ float ldT(const void* p, int64_t i) { return Elem<T>::ld((const T*)p + i); }
template <typename T> __device__ __forceinline__ void stT(void* p, int64_t i, float v) { Elem<T>::st((T*)p + i, v); }

// ---- pack_weight ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pack_weight_body(const ZsPackWeight& p, int64_t first, int64_t step) {
  const int64_t total = (int64_t)p.n_rows * p.n_cols;
  for (int64_t i = first; i < total; i += step) {
    const int row = (int)(i / p.n_cols);
    const int k = (int)(i - (int64_t)row * p.n_cols);
    const int tap = k / p.inner_pad, inner = k - tap * p.inner_pad;
    int n, ci;
    if (p.transpose == 0) { n = row; ci = inner; } else { n = inner; ci = row; }
    float v = 0.f;
    if (n < p.Cout && ci < p.Cin && tap < p.taps) {
      const int half = p.Cout >> 1;
      const int co = p.row_perm ? p.row_perm[n] : (p.co_split2 ? (n < half ? 2 * n : 2 * (n - half) + 1) : n);
      v = p.W[(int64_t)co * p.so + (int64_t)ci * p.si + (int64_t)tap * p.sj];
    }
    stT<T>(p.dst, (int64_t)(row + p.row_offset) * p.ldw + p.col_offset + k, v);
  }
}

template <typename T>
__global__ void pack_weight_kernel(const ZsPackWeight p) {
  pack_weight_body<T>(p, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// Tiled variant (taps <= PK_MAX_TAPS): a block moves a tile of R destination rows x I inner positions x all taps through LDS,
// so that both sides are coalesced -- the source run that is contiguous in memory is (ci, tap) for a fixed output channel, the
// destination run is the inner index.  transpose 0: rows = n, inner = ci (R x 128); transpose 1: rows = ci, inner = n (R x 64);
// R is the power of two that keeps the tile within 16 KiB of LDS (8 blocks per CU: the kernel is latency-bound otherwise).
// The element-per-thread kernel above reads the transposed layout with a stride of Cin*k floats per lane (1.1 TB/s).
constexpr int PK_THREADS = 256, PK_MAX_TAPS = 7;
__host__ __device__ constexpr int pk_pow2_le(int x) { return x >= 64 ? 64 : x >= 32 ? 32 : x >= 16 ? 16 : x >= 8 ? 8 : x >= 4 ? 4 : x >= 2 ? 2 : 1; }
__host__ __device__ constexpr int pk_inner(int transpose) { return transpose ? 64 : 128; }
__host__ __device__ constexpr int pk_rows(int transpose, int taps) { return pk_pow2_le((transpose ? 64 : 32) / taps); }
inline int pk_tiles(const ZsPackWeight& p) {
  const int R = pk_rows(p.transpose, p.taps), I = pk_inner(p.transpose);
  return ((p.n_rows + R - 1) / R) * ((p.inner_pad + I - 1) / I);
}
constexpr size_t PK_LDS_BYTES = (4096 + 128) * sizeof(float);

template <typename T, int TAPS, int TR>
__device__ __forceinline__ void pack_tile(const ZsPackWeight& p, int tile, float* lds) {
  constexpr int R = pk_rows(TR, TAPS), I = pk_inner(TR);
  const int tid = threadIdx.x;
  const int tiles_i = (p.inner_pad + I - 1) / I;
  const int tr = tile / tiles_i, ti = tile - tr * tiles_i;
  const int r0 = tr * R, i0 = ti * I;
  if (r0 >= p.n_rows) return;
  const int half = p.Cout >> 1;
  auto co_of = [&](int n) { return p.row_perm ? p.row_perm[n] : (p.co_split2 ? (n < half ? 2 * n : 2 * (n - half) + 1) : n); };
  if (TR == 0) {
    constexpr int RUN = I * TAPS, PITCH = RUN + 1;            // lds[rl][il * TAPS + tap]
#pragma unroll 4
    for (int e = tid; e < R * RUN; e += PK_THREADS) {
      const int rl = e / RUN, x = e - rl * RUN;
      const int il = x / TAPS, tap = x - il * TAPS;
      const int n = r0 + rl, ci = i0 + il;
      float v = 0.f;
      if (n < p.Cout && ci < p.Cin) v = p.W[(int64_t)co_of(n) * p.so + (int64_t)ci * p.si + (int64_t)tap * p.sj];
      lds[rl * PITCH + x] = v;
    }
  } else {
    constexpr int RUN = R * TAPS, PITCH = RUN + 1;            // lds[il][rl * TAPS + tap]
#pragma unroll 4
    for (int e = tid; e < I * RUN; e += PK_THREADS) {
      const int il = e / RUN, x = e - il * RUN;
      const int rl = x / TAPS, tap = x - rl * TAPS;
      const int n = i0 + il, ci = r0 + rl;
      float v = 0.f;
      if (n < p.Cout && ci < p.Cin) v = p.W[(int64_t)co_of(n) * p.so + (int64_t)ci * p.si + (int64_t)tap * p.sj];
      lds[il * PITCH + x] = v;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int e = tid; e < R * TAPS * I; e += PK_THREADS) {
    const int il = e % I, t2 = e / I;
    const int tap = t2 % TAPS, rl = t2 / TAPS;
    const int row = r0 + rl, inner = i0 + il;
    if (row < p.n_rows && inner < p.inner_pad) {
      const float v = TR ? lds[il * (R * TAPS + 1) + rl * TAPS + tap] : lds[rl * (I * TAPS + 1) + il * TAPS + tap];
      stT<T>(p.dst, (int64_t)(row + p.row_offset) * p.ldw + p.col_offset + tap * p.inner_pad + inner, v);
    }
  }
  const int tail0 = TAPS * p.inner_pad, tw = p.n_cols - tail0;      // columns past the last tap: zeros
  if (ti == 0 && tw > 0) {
    for (int e = tid; e < R * tw; e += PK_THREADS) {
      const int rl = e / tw, k = e - rl * tw;
      if (r0 + rl < p.n_rows) stT<T>(p.dst, (int64_t)(r0 + rl + p.row_offset) * p.ldw + p.col_offset + tail0 + k, 0.f);
    }
  }
}

template <typename T>
__device__ __forceinline__ void pack_tile_any(const ZsPackWeight& p, int tile, float* lds) {
#define ZS_PK_CASE(K) case K: if (p.transpose) pack_tile<T, K, 1>(p, tile, lds); else pack_tile<T, K, 0>(p, tile, lds); break;
  switch (p.taps) { ZS_PK_CASE(1) ZS_PK_CASE(2) ZS_PK_CASE(3) ZS_PK_CASE(4) ZS_PK_CASE(5) ZS_PK_CASE(6) ZS_PK_CASE(7) default: break; }
#undef ZS_PK_CASE
}

template <typename T>
__global__ __launch_bounds__(PK_THREADS) void pack_weight_tiled_kernel(const ZsPackWeight p) {
  extern __shared__ float pk_lds[];
  pack_tile_any<T>(p, blockIdx.x, pk_lds);
}

// up to PACK_BATCH jobs per launch (the per-step repack of a net is ~40 small jobs: one launch instead of 40)
constexpr int PACK_BATCH = 32;
struct PackBatch { ZsPackWeight job[PACK_BATCH]; int32_t tile_start[PACK_BATCH + 1]; int32_t n; };

// one flat tile list over the jobs (tile_start = prefix sums); the workgroups walk it with the grid as stride
template <typename T>
__global__ __launch_bounds__(PK_THREADS) void pack_weight_batch_kernel(const PackBatch b) {
  extern __shared__ float pk_lds[];
  const int total = b.tile_start[b.n];
  for (int g = blockIdx.x; g < total; g += gridDim.x) {
    int j = 0;
    while (g >= b.tile_start[j + 1]) ++j;                                  // (block-uniform)
    const ZsPackWeight& p = b.job[j];
    const int tile = g - b.tile_start[j], ntile = b.tile_start[j + 1] - b.tile_start[j];
    if (p.taps <= PK_MAX_TAPS) pack_tile_any<T>(p, tile, pk_lds);
    else pack_weight_body<T>(p, (int64_t)tile * blockDim.x + threadIdx.x, (int64_t)ntile * blockDim.x);
    __syncthreads();
  }
}

// ---- small vector copies, one launch for all -------------------------------------------------------
constexpr int VEC_BATCH = 32;
struct VecBatch { ZsVecCopy job[VEC_BATCH]; };
__global__ void copy_vec_batch_kernel(const VecBatch b) {
  const ZsVecCopy& j = b.job[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < j.len; i += gridDim.x * blockDim.x)
    j.dst[(int64_t)i * j.dst_stride] = j.src[(int64_t)i * j.src_stride];
}

// ---- cast_rows -------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_rows_kernel(const ZsCastRows p) {
  // one wave per row at a time (no per-element 64-bit division: the element-indexed form ran at 2.8 TB/s); a lane takes a PAIR of
  // columns, so that the bf16 destinations get 4-byte stores where their offsets allow it
  const int fc = (p.dst2 && p.fill_cols2 > p.fill_cols) ? p.fill_cols2 : p.fill_cols;
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * wpb;
  const bool pair1 = !p.dst_f32 && sizeof(T) == 2 && ((p.col_off | (int)(p.ld_dst & 1)) & 1) == 0 && (((uintptr_t)p.dst) & 3) == 0;
  const bool pair2 = p.dst2 && sizeof(T) == 2 && ((p.col_off2 | (int)(p.ld_dst2 & 1)) & 1) == 0 && (((uintptr_t)p.dst2) & 3) == 0;
  for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < p.rows; r += nw) {
    for (int c = 2 * lane; c < fc; c += 128) {
      float x[2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        x[e] = (c + e < p.cols) ? (p.src_f32 ? ((const float*)p.src)[r * p.ld_src + c + e] : ldT<T>(p.src, r * p.ld_src + c + e)) : 0.f;
      {
        const float v0 = (p.act == ZS_ACT_LRELU) ? lrelu_f(x[0], p.slope) : x[0], v1 = (p.act == ZS_ACT_LRELU) ? lrelu_f(x[1], p.slope) : x[1];
        if (pair1 && c + 1 < p.fill_cols) {
          *reinterpret_cast<uint32_t*>((bf16_t*)p.dst + r * p.ld_dst + p.col_off + c) = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
        } else {
          if (c < p.fill_cols) { if (p.dst_f32) ((float*)p.dst)[r * p.ld_dst + p.col_off + c] = v0; else stT<T>(p.dst, r * p.ld_dst + p.col_off + c, v0); }
          if (c + 1 < p.fill_cols) { if (p.dst_f32) ((float*)p.dst)[r * p.ld_dst + p.col_off + c + 1] = v1; else stT<T>(p.dst, r * p.ld_dst + p.col_off + c + 1, v1); }
        }
      }
      if (p.dst2) {
        const float v0 = (p.act2 == ZS_ACT_LRELU) ? lrelu_f(x[0], p.slope2) : x[0], v1 = (p.act2 == ZS_ACT_LRELU) ? lrelu_f(x[1], p.slope2) : x[1];
        if (pair2 && c + 1 < p.fill_cols2) {
          *reinterpret_cast<uint32_t*>((bf16_t*)p.dst2 + r * p.ld_dst2 + p.col_off2 + c) = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
        } else {
          if (c < p.fill_cols2) stT<T>(p.dst2, r * p.ld_dst2 + p.col_off2 + c, v0);
          if (c + 1 < p.fill_cols2) stT<T>(p.dst2, r * p.ld_dst2 + p.col_off2 + c + 1, v1);
        }
      }
    }
  }
}

// ---- add_rowvec ------------------------------------------------------------------------------------
template <typename T>
__global__ void add_rowvec_kernel(const ZsAddRowvec p) {
  const int64_t rows = (int64_t)p.B * p.T;
  const int64_t total = rows * p.fill_cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / p.fill_cols;
    const int c = (int)(i - r * p.fill_cols);
    const int b = (int)(r / p.T);
    float v = p.x ? ldT<T>(p.x, r * p.ldx + c) : 0.f;
    if (c < p.C) v += p.vec[p.idx[b] * p.vec_ld + c];
    stT<T>(p.out, r * p.ldo + c, v);
  }
}

// broadcast form (x == null, append_emb): 8 columns = one 16-byte store per lane
template <typename T>
__global__ void add_rowvec_bcast8_kernel(const ZsAddRowvec p) {
  const int groups = p.fill_cols / 8;
  const int64_t total = (int64_t)p.B * p.T * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / groups;
    const int c0 = (int)(i - r * groups) * 8;
    const int b = (int)(r / p.T);
    const float* v = p.vec + p.idx[b] * p.vec_ld + c0;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (c0 + e < p.C) ? v[e] : 0.f;
    store8<T>((T*)p.out + r * p.ldo + c0, o);
  }
}

// ---- embedding scatter (fixed sample order, no atomics) ---------------------------------------------
// grid (C / 128, embedding rows), 128 threads: the samples of row r are found 128 at a time (ballot + prefix into an LDS list, in
// ascending sample order) and only those are summed -- the plain loop over all B samples with a compare each was 25 us of
// latency for a few KB of data, five times per step on the chain.
__global__ __launch_bounds__(128) void emb_scatter_kernel(const ZsEmbScatter p) {
  __shared__ int list[128];
  __shared__ int wcount[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x * 128 + tid;
  const int r = blockIdx.y;
  float s = 0.f;
  for (int b0 = 0; b0 < p.B; b0 += 128) {
    const int b = b0 + tid;
    const bool hit = b < p.B && p.idx[b] == (int64_t)r;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    const int n0 = wcount[0], n = n0 + wcount[1];
    if (hit) list[(wave ? n0 : 0) + __popcll(m & ((1ull << lane) - 1ull))] = b;
    __syncthreads();
    if (c < p.C)
      for (int i = 0; i < n; ++i) s += p.emb_sum[(int64_t)list[i] * p.emb_ld + c];
    __syncthreads();
  }
  if (c >= p.C) return;
  float* d = p.demb + (int64_t)r * p.demb_ld + c;
  *d = p.accumulate ? (*d + s) : s;
}

// ---- MBV ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void mbv_fwd_kernel(const ZsMbvFwd p) {
  const int64_t total = p.rows * p.bits_fill_cols;
  const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / p.bits_fill_cols;
    const int e = (int)(i - r * p.bits_fill_cols);
    float bit = 0.f;
    if (e < p.E) {
      float l0, l1;
      if (p.logits_f32) { l0 = ((const float*)p.logits)[r * p.ld + 2 * e]; l1 = ((const float*)p.logits)[r * p.ld + 2 * e + 1]; }
      else { l0 = ldT<T>(p.logits, r * p.ld + 2 * e); l1 = ldT<T>(p.logits, r * p.ld + 2 * e + 1); }
      const int64_t ni = (r * p.E + e) * 2;
      float g0, g1;
      if (p.noise_kind == 0) { g0 = p.noise[ni]; g1 = p.noise[ni + 1]; }
      else {
        float u0, u1;
        if (p.noise_kind == 1) { u0 = p.noise[ni]; u1 = p.noise[ni + 1]; }
        else { u0 = zs_uniform(seed, 0x6d627600u, (uint64_t)ni); u1 = zs_uniform(seed, 0x6d627600u, (uint64_t)ni + 1); }
        g0 = -logf(-logf(u0 + 1e-20f) + 1e-20f);                 // model/model.py:95-98
        g1 = -logf(-logf(u1 + 1e-20f) + 1e-20f);
      }
      const float a0 = __fdiv_rn(__fadd_rn(l0, g0), p.tau);     // (logits + G) / temperature, true fp32 division
      const float a1 = __fdiv_rn(__fadd_rn(l1, g1), p.tau);
      const float mx = fmaxf(a0, a1);
      const float e0 = expf(a0 - mx), e1 = expf(a1 - mx);
      const float sum = __fadd_rn(e0, e1);
      const float y0 = __fdiv_rn(e0, sum), y1 = __fdiv_rn(e1, sum);
      const float hard = (y0 >= y1) ? 1.f : 0.f;                 // argmax, first index on ties
      bit = __fadd_rn(__fsub_rn(hard, y0), y0);                  // (y_hard - y).detach() + y
      if (p.bits_f32) p.bits_f32[r * p.E + e] = bit;
      if (p.y0) p.y0[r * p.E + e] = y0;
    }
    if (p.bits) stT<T>(p.bits, r * p.ld_bits + e, bit);
  }
}

template <typename T>
__global__ void mbv_bwd_kernel(const ZsMbvBwd p) {
  const int64_t total = p.rows * (p.fill_cols / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (p.fill_cols / 2);
    const int e = (int)(i - r * (p.fill_cols / 2));
    float d0 = 0.f;
    if (e < p.E) {
      const float y0 = p.y0[r * p.E + e];
      d0 = ldT<T>(p.dbits, r * p.ld_dbits + e) * y0 * (1.f - y0) / p.tau;
    }
    stT<T>(p.dlogits, r * p.ld + 2 * e, d0);
    stT<T>(p.dlogits, r * p.ld + 2 * e + 1, -d0);
  }
}

// ---- L1 loss -----------------------------------------------------------------------------------------
template <typename T>
__global__ void l1_stage1_kernel(const ZsL1Loss p) {
  __shared__ float red[NTE / 64];
  const float gs = p.grad_scale / ((float)p.rows * (float)p.F);
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * wpb;
  float s = 0.f;
  const bool pair = p.dlogits && sizeof(T) == 2 && (p.ldg & 1) == 0 && (((uintptr_t)p.dlogits) & 3) == 0;
  for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < p.rows; r += nw) {      // one wave per row at a time
    for (int c = 2 * lane; c < p.fill_cols; c += 128) {                                         // a lane takes a pair of columns
      float g[2] = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (c + e < p.F) {
          const float xd = p.x_dec[r * p.ld_dec + c + e];
          const float d = xd - p.x[r * p.ldx + c + e];
          s += fabsf(d);
          g[e] = (d > 0.f ? gs : (d < 0.f ? -gs : 0.f)) * xd * (1.f - xd);
        }
      if (pair && c + 1 < p.fill_cols) {
        *reinterpret_cast<uint32_t*>((bf16_t*)p.dlogits + r * p.ldg + c) = (uint32_t)f2bf(g[0]) | ((uint32_t)f2bf(g[1]) << 16);
      } else if (p.dlogits) {
        stT<T>(p.dlogits, r * p.ldg + c, g[0]);
        if (c + 1 < p.fill_cols) stT<T>(p.dlogits, r * p.ldg + c + 1, g[1]);
      }
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < NTE / 64; ++w) t += red[w]; p.partial[blockIdx.x] = t; }
}
__global__ void l1_stage2_kernel(const float* partial, int n, float* out, double denom) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
}

// ---- squared norm --------------------------------------------------------------------------------------
__global__ void sqnorm_stage1_kernel(const float* g, int64_t n, double* partial) {
  __shared__ double red[NTE / 64];
  double s = 0.0;
  const int64_t n4 = ((((uintptr_t)g) & 15) == 0) ? (n >> 2) : 0;          // 16-byte loads, two in flight per lane
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 a = g4[i], b = g4[i + stride];
    s += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    s += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
  }
  for (; i < n4; i += stride) {
    const float4 a = g4[i];
    s += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
  }
  for (int64_t k = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const float v = g[k]; s += (double)v * (double)v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NTE / 64; ++w) t += red[w]; partial[blockIdx.x] = t; }
}
__global__ void sqnorm_stage2_kernel(const double* partial, int n, float* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---- Adam + clip -------------------------------------------------------------------------------------
__global__ void adam_kernel(const ZsAdam p) {
  const float gs = p.grad_scale != 0.f ? p.grad_scale : 1.f;      // 1/world: g holds the SUM over the data-parallel ranks
  float coef = gs;
  if (p.max_norm > 0.f && p.sumsq) {
    const float norm = sqrtf(*p.sumsq) * gs;
    coef = gs * fminf(p.max_norm / (norm + 1e-6f), 1.f);         // clip_grad_norm_
  }
  float bc1 = p.bc1, bc2 = p.bc2;
  if (p.step_ptr) {                                              // hipGraph replay: the step count lives on the device
    const float t = (float)(*p.step_ptr);
    bc1 = 1.f - powf(p.beta1, t);
    bc2 = 1.f - powf(p.beta2, t);
  }
  const float step_size = p.lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  auto upd = [&](float& w, float& g, float& m, float& v) {
    g *= coef;
    m = m + (g - m) * (1.f - p.beta1);                           // exp_avg.lerp_(grad, 1 - beta1)
    v = v * p.beta2 + g * g * (1.f - p.beta2);                   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / bc2_sqrt + p.eps;
    w = w - step_size * (m / denom);
  };
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  const bool vec = ((((uintptr_t)p.p) | ((uintptr_t)p.g) | ((uintptr_t)p.m) | ((uintptr_t)p.v)) & 15) == 0;
  const int64_t n4 = vec ? p.n / 4 : 0;
  for (int64_t i = tid; i < n4; i += nthr) {                     // 16-byte accesses: four streams of 64 B per lane in flight
    float4 w = reinterpret_cast<float4*>(p.p)[i], g = reinterpret_cast<float4*>(p.g)[i];
    float4 m = reinterpret_cast<float4*>(p.m)[i], v = reinterpret_cast<float4*>(p.v)[i];
    upd(w.x, g.x, m.x, v.x); upd(w.y, g.y, m.y, v.y); upd(w.z, g.z, m.z, v.z); upd(w.w, g.w, m.w, v.w);
    reinterpret_cast<float4*>(p.p)[i] = w; reinterpret_cast<float4*>(p.m)[i] = m; reinterpret_cast<float4*>(p.v)[i] = v;
    if (p.write_clipped_grad) reinterpret_cast<float4*>(p.g)[i] = g;
  }
  for (int64_t i = 4 * n4 + tid; i < p.n; i += nthr) {
    float w = p.p[i], g = p.g[i], m = p.m[i], v = p.v[i];
    upd(w, g, m, v);
    p.p[i] = w; p.m[i] = m; p.v[i] = v;
    if (p.write_clipped_grad) p.g[i] = g;
  }
}

// pinned host memory -> device, 8 x 16-byte loads in flight per lane (PCIe reads: latency ~2 us, 50+ GB/s)
typedef __attribute__((ext_vector_type(4))) unsigned fetch_u32x4;
__global__ __launch_bounds__(256) void host_fetch_kernel(const fetch_u32x4* __restrict__ src, fetch_u32x4* __restrict__ dst, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    fetch_u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

__global__ void step_counters_kernel(uint64_t* seed, int32_t* step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (seed) *seed += 0x9E3779B97F4A7C15ull;
    if (step) *step += 1;
  }
}

// ---- softmax cross-entropy ----------------------------------------------------------------------------
__global__ void softmax_ce_kernel(const ZsSoftmaxCE p) {
  __shared__ float red[4];
  __shared__ int redc[4];
  float ls = 0.f; int corr = 0;
  for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
    const float* l = p.logits + (int64_t)b * p.ld;
    float mx = l[0]; int am = 0;
    for (int j = 1; j < p.n_class; ++j) if (l[j] > mx) { mx = l[j]; am = j; }
    float se = 0.f;
    for (int j = 0; j < p.n_class; ++j) se += expf(l[j] - mx);
    const int t = (int)p.target[b];
    ls += (logf(se) + mx) - l[t];
    corr += (am == t);
    if (p.dlogits) {
      const float gs = p.grad_scale / (float)p.B;
      for (int j = 0; j < p.n_class; ++j)
        p.dlogits[(int64_t)b * p.ldg + j] = (expf(l[j] - mx) / se - (j == t ? 1.f : 0.f)) * gs;
    }
  }
  ls = wave_sum(ls);
  corr = (int)wave_sum((float)corr);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ls; redc[threadIdx.x >> 6] = corr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    *p.loss_out = (red[0] + red[1] + red[2] + red[3]) / (float)p.B;
    if (p.correct_out) *p.correct_out = redc[0] + redc[1] + redc[2] + redc[3];
  }
}

inline unsigned nblocks(int64_t total, int cap = 2048) {
  int64_t b = (total + NTE - 1) / NTE;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

#define ZS_DISPATCH(dtype, KERNEL, grid, block, stream, ...)                                            \
  do {                                                                                                  \
    if ((dtype) == ZS_F32) hipLaunchKernelGGL(KERNEL<float>, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__);          \
  } while (0)

// in-place per-sample row reversal over a column block (zs_rows_reverse): one thread per (sample, row pair, 4-byte word)
__global__ __launch_bounds__(NTE) void rows_reverse_kernel(const ZsRowsReverse p, int words, int64_t ld_bytes, int64_t col_bytes) {
  const int half = p.T >> 1;
  const int64_t total = (int64_t)p.B * half * words;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % words);
    const int64_t r = i / words;
    const int t = (int)(r % half), b = (int)(r / half);
    const int len = p.lengths ? p.lengths[b] : p.T;
    if (t >= (len >> 1)) continue;
    unsigned char* base = (unsigned char*)p.x + col_bytes + (int64_t)w * 4;
    uint32_t* lo = reinterpret_cast<uint32_t*>(base + ((int64_t)b * p.T + t) * ld_bytes);
    uint32_t* hi = reinterpret_cast<uint32_t*>(base + ((int64_t)b * p.T + (len - 1 - t)) * ld_bytes);
    const uint32_t a = *lo, c = *hi;
    *lo = c; *hi = a;
  }
}

static int pack_check(const ZsPackWeight* p) {
  ZS_REQUIRE(p && p->W && p->dst, "zs_pack_weight: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_pack_weight: bad dtype");
  ZS_REQUIRE(p->Cout > 0 && p->Cin > 0 && p->taps > 0 && p->inner_pad > 0 && p->n_rows > 0 && p->n_cols > 0 &&
                 p->ldw >= (int64_t)p->col_offset + p->n_cols && p->n_cols >= p->taps * p->inner_pad,
             "zs_pack_weight: sizes (ldw %lld n_cols %d taps %d inner_pad %d)", (long long)p->ldw, p->n_cols, p->taps, p->inner_pad);
  ZS_REQUIRE((p->transpose ? p->Cout : p->Cin) <= p->inner_pad, "zs_pack_weight: inner_pad too small");
  ZS_REQUIRE(!p->co_split2 || p->Cout % 2 == 0, "zs_pack_weight: SPLIT2 needs even Cout");
  return ZS_OK;
}

extern "C" int zs_pack_weight(const ZsPackWeight* p, void* stream) {
  int rc = pack_check(p);
  if (rc) return rc;
  if (p->taps <= PK_MAX_TAPS) {
    const size_t lds = PK_LDS_BYTES;
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(pack_weight_tiled_kernel<float>, dim3(pk_tiles(*p)), dim3(PK_THREADS), lds, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(pack_weight_tiled_kernel<bf16_t>, dim3(pk_tiles(*p)), dim3(PK_THREADS), lds, (hipStream_t)stream, *p);
  } else {
    ZS_DISPATCH(p->dtype, pack_weight_kernel, dim3(nblocks((int64_t)p->n_rows * p->n_cols, 4096)), dim3(NTE), stream, *p);
  }
  return zs_check_launch("zs_pack_weight");
}

extern "C" int zs_pack_weight_batch(const ZsPackWeight* jobs, int32_t n, int32_t max_blocks, void* stream) {
  ZS_REQUIRE(jobs && n > 0, "zs_pack_weight_batch: no jobs");
  for (int i = 0; i < n; ++i) {
    int rc = pack_check(jobs + i);
    if (rc) return rc;
    ZS_REQUIRE(jobs[i].dtype == jobs[0].dtype, "zs_pack_weight_batch: mixed dtypes");
  }
  for (int i0 = 0; i0 < n; i0 += PACK_BATCH) {
    const int m = n - i0 < PACK_BATCH ? n - i0 : PACK_BATCH;
    PackBatch b;
    memset(&b, 0, sizeof(b));
    int64_t total = 0;
    for (int i = 0; i < m; ++i) {
      b.job[i] = jobs[i0 + i];
      const ZsPackWeight& j = jobs[i0 + i];
      b.tile_start[i] = (int32_t)total;
      total += j.taps <= PK_MAX_TAPS ? (int64_t)pk_tiles(j) : (int64_t)nblocks((int64_t)j.n_rows * j.n_cols, 4096);
      ZS_REQUIRE(total < (1ll << 30), "zs_pack_weight_batch: too many tiles");
    }
    for (int i = m; i <= PACK_BATCH; ++i) b.tile_start[i] = (int32_t)total;
    b.n = m;
    const unsigned gx = (unsigned)((max_blocks > 0 && total > max_blocks) ? max_blocks : total);
    const size_t lds = PK_LDS_BYTES;
    if (jobs[0].dtype == ZS_F32) hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(gx), dim3(PK_THREADS), lds, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(pack_weight_batch_kernel<bf16_t>, dim3(gx), dim3(PK_THREADS), lds, (hipStream_t)stream, b);
    int rc = zs_check_launch("zs_pack_weight_batch");
    if (rc) return rc;
  }
  return ZS_OK;
}

extern "C" int zs_copy_vec_batch(const ZsVecCopy* jobs, int32_t n, void* stream) {
  ZS_REQUIRE(jobs && n > 0, "zs_copy_vec_batch: no jobs");
  for (int i = 0; i < n; ++i)
    ZS_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].len > 0, "zs_copy_vec_batch: bad job %d", i);
  for (int i0 = 0; i0 < n; i0 += VEC_BATCH) {
    const int m = n - i0 < VEC_BATCH ? n - i0 : VEC_BATCH;
    VecBatch b;
    memset(&b, 0, sizeof(b));
    int mx = 1;
    for (int i = 0; i < m; ++i) { b.job[i] = jobs[i0 + i]; if (jobs[i0 + i].len > mx) mx = jobs[i0 + i].len; }
    unsigned gx = (unsigned)((mx + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_vec_batch_kernel, dim3(gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, b);
    int rc = zs_check_launch("zs_copy_vec_batch");
    if (rc) return rc;
  }
  return ZS_OK;
}

extern "C" int zs_cast_rows(const ZsCastRows* p, void* stream) {
  ZS_REQUIRE(p && p->src && p->dst && p->rows > 0 && p->cols > 0 && p->fill_cols >= p->cols, "zs_cast_rows: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_cast_rows: bad dtype");
  ZS_REQUIRE(!p->dst2 || p->fill_cols2 >= p->cols, "zs_cast_rows: fill_cols2 < cols");
  ZS_DISPATCH(p->dtype, cast_rows_kernel, dim3(nblocks(p->rows * 64, 2048)), dim3(NTE), stream, *p);
  return zs_check_launch("zs_cast_rows");
}

extern "C" int zs_add_rowvec(const ZsAddRowvec* p, void* stream) {
  ZS_REQUIRE(p && p->vec && p->idx && p->out && p->B > 0 && p->T > 0 && p->C > 0 && p->fill_cols >= p->C, "zs_add_rowvec: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_add_rowvec: bad dtype");
  if (!p->x && p->fill_cols % 8 == 0 && p->ldo % 8 == 0 && (((uintptr_t)p->out) & 15) == 0) {
    ZS_DISPATCH(p->dtype, add_rowvec_bcast8_kernel, dim3(nblocks((int64_t)p->B * p->T * (p->fill_cols / 8), 4096)), dim3(NTE), stream, *p);
    return zs_check_launch("zs_add_rowvec");
  }
  ZS_DISPATCH(p->dtype, add_rowvec_kernel, dim3(nblocks((int64_t)p->B * p->T * p->fill_cols, 4096)), dim3(NTE), stream, *p);
  return zs_check_launch("zs_add_rowvec");
}

extern "C" int zs_rows_reverse(const ZsRowsReverse* p, void* stream) {
  ZS_REQUIRE(p && p->x && p->B > 0 && p->T > 0 && p->cols > 0 && p->col0 >= 0 && p->ld >= (int64_t)p->col0 + p->cols, "zs_rows_reverse: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_rows_reverse: bad dtype");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE((p->cols * es) % 4 == 0 && (p->col0 * es) % 4 == 0 && (p->ld * es) % 4 == 0 && (((uintptr_t)p->x) & 3) == 0,
             "zs_rows_reverse: the column block must be made of whole 4-byte words");
  if (p->T < 2) return ZS_OK;
  const int words = p->cols * es / 4;
  hipLaunchKernelGGL(rows_reverse_kernel, dim3(nblocks((int64_t)p->B * (p->T >> 1) * words, 4096)), dim3(NTE), 0, (hipStream_t)stream, *p, words,
                     (int64_t)p->ld * es, (int64_t)p->col0 * es);
  return zs_check_launch("zs_rows_reverse");
}

extern "C" int zs_emb_scatter(const ZsEmbScatter* p, void* stream) {
  ZS_REQUIRE(p && p->emb_sum && p->idx && p->demb && p->B > 0 && p->n_rows > 0 && p->C > 0, "zs_emb_scatter: bad args");
  hipLaunchKernelGGL(emb_scatter_kernel, dim3((p->C + 127) / 128, p->n_rows), dim3(128), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_emb_scatter");
}

extern "C" int zs_mbv_fwd(const ZsMbvFwd* p, void* stream) {
  ZS_REQUIRE(p && p->logits && p->rows > 0 && p->E > 0 && p->tau > 0.f, "zs_mbv_fwd: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_mbv_fwd: bad dtype");
  ZS_REQUIRE(p->noise_kind == 2 || p->noise, "zs_mbv_fwd: noise missing");
  ZS_REQUIRE(p->bits || p->bits_f32, "zs_mbv_fwd: no output");
  ZsMbvFwd q = *p;
  if (!q.bits || q.bits_fill_cols < q.E) q.bits_fill_cols = q.E;
  ZS_DISPATCH(p->dtype, mbv_fwd_kernel, dim3(nblocks(q.rows * q.bits_fill_cols)), dim3(NTE), stream, q);
  return zs_check_launch("zs_mbv_fwd");
}

extern "C" int zs_mbv_bwd(const ZsMbvBwd* p, void* stream) {
  ZS_REQUIRE(p && p->dbits && p->y0 && p->dlogits && p->rows > 0 && p->E > 0 && p->fill_cols >= 2 * p->E && p->fill_cols % 2 == 0,
             "zs_mbv_bwd: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_mbv_bwd: bad dtype");
  ZS_DISPATCH(p->dtype, mbv_bwd_kernel, dim3(nblocks(p->rows * (p->fill_cols / 2))), dim3(NTE), stream, *p);
  return zs_check_launch("zs_mbv_bwd");
}

extern "C" int zs_l1_loss(const ZsL1Loss* p, void* stream) {
  ZS_REQUIRE(p && p->x_dec && p->x && p->partial && p->loss_out && p->rows > 0 && p->F > 0, "zs_l1_loss: bad args");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_l1_loss: bad dtype");
  ZsL1Loss q = *p;
  if (!q.dlogits || q.fill_cols < q.F) q.fill_cols = q.F;
  const unsigned nb = nblocks(q.rows * 64, 1024);
  ZS_DISPATCH(p->dtype, l1_stage1_kernel, dim3(nb), dim3(NTE), stream, q);
  int rc = zs_check_launch("zs_l1_loss.stage1");
  if (rc) return rc;
  hipLaunchKernelGGL(l1_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)p->partial, (int)nb, p->loss_out,
                     (double)p->rows * (double)p->F);
  return zs_check_launch("zs_l1_loss.stage2");
}

extern "C" int zs_sqnorm(const float* g, int64_t n, double* partial, float* out_sq, void* stream) {
  ZS_REQUIRE(g && partial && out_sq && n > 0, "zs_sqnorm: bad args");
  const unsigned nb = nblocks(n, 1024);
  hipLaunchKernelGGL(sqnorm_stage1_kernel, dim3(nb), dim3(NTE), 0, (hipStream_t)stream, g, n, partial);
  int rc = zs_check_launch("zs_sqnorm.stage1");
  if (rc) return rc;
  hipLaunchKernelGGL(sqnorm_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial, (int)nb, out_sq);
  return zs_check_launch("zs_sqnorm.stage2");
}

extern "C" int zs_adam_clip(const ZsAdam* p, void* stream) {
  ZS_REQUIRE(p && p->p && p->g && p->m && p->v && p->n > 0, "zs_adam_clip: bad args");
  ZS_REQUIRE(p->step_ptr || (p->bc1 > 0.f && p->bc2 > 0.f), "zs_adam_clip: bias corrections");
  unsigned nb = nblocks((p->n + 3) / 4, 4096);
  if (p->max_blocks > 0 && nb > (unsigned)p->max_blocks) nb = (unsigned)p->max_blocks;
  hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(NTE), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_adam_clip");
}

extern "C" int zs_host_fetch(const void* src, void* dst, size_t bytes, int32_t workgroups, void* stream) {
  ZS_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, "zs_host_fetch: bad args (16-byte granularity)");
  const int wg = workgroups > 0 ? workgroups : 32;
  hipLaunchKernelGGL(host_fetch_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, (const fetch_u32x4*)src, (fetch_u32x4*)dst, (int64_t)(bytes / 16));
  return zs_check_launch("zs_host_fetch");
}

extern "C" int zs_step_counters(uint64_t* seed, int32_t* step, void* stream) {
  ZS_REQUIRE(seed || step, "zs_step_counters: bad args");
  hipLaunchKernelGGL(step_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed, step);
  return zs_check_launch("zs_step_counters");
}

extern "C" int zs_softmax_ce(const ZsSoftmaxCE* p, void* stream) {
  ZS_REQUIRE(p && p->logits && p->target && p->loss_out && p->B > 0 && p->n_class > 0, "zs_softmax_ce: bad args");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_softmax_ce");
}
