// zs_patch.hip -- stage 2 (PatchDiscriminator / TargetClassifier, WGAN-GP): the pieces around the implicit-GEMM kernels that a
// 2-D convolution stack needs (reference model/model.py:113-228, utils.py:58-77, trainer.py:257-294).
//
//  * Conv2d k x k / stride s on channels-last [B, H, W, C]: im2col along H only (zs_conv2d_gather, k/s times the input), then the
//    existing zs_gemm_conv over W -- taps along W, reflect / zero padding and stride in its row gather, "batch" = (b, ho) --
//    on the weight viewed as a Conv1d weight [Cout, k*C, k].  The data gradient comes back from zs_gemm_conv(gather=1) in the
//    W-padded domain; zs_conv2d_fold applies the reflect folds along W and the transpose of the H gather as ONE gather-formulated
//    sum per output element (fixed order, no atomics).
//  * InstanceNorm2d + Dropout2d over T = H*W up to 16 448 rows per sample: statistics are two-stage column reductions
//    (zs_row_moments: 512-row slabs -> partials -> fixed-order finish), the normalisation, its backward and the double backward
//    of the gradient penalty are row-streaming elementwise kernels with per-(b, c) coefficients.  HBM-bound, 16-byte accesses.
#include "zs_common.h"

namespace {

constexpr int NTP = 256;
constexpr int MOM_ROWS = 512;      // rows of one partial slab of zs_row_moments

__device__ __forceinline__ int pad_index(int s, int n, int mode, bool& valid) {
  valid = true;
  if (s >= 0 && s < n) return s;
  if (mode == ZS_PAD_REFLECT) return zs_reflect(s, n);
  valid = false;
  return 0;
}

// ---- Conv2d: im2col along H ------------------------------------------------------------------------------------------------
// one thread per (output row (b, ho, w), 8-element group of the k*C output columns): 16-byte stores; source elements of a
// group may straddle two kh blocks when C is not a multiple of 8 (C = 1: the first layer), so the general path is per element.
template <typename T>
__global__ __launch_bounds__(NTP) void conv2d_gather_kernel(const ZsConv2dGather p) {
  const int groups = (int)(p.ldo / 8);
  const int64_t total = (int64_t)p.B * p.H_out * p.Wd * groups;
  const int kc = p.k * p.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const int64_t row = i / groups;
    const int w = (int)(row % p.Wd);
    const int64_t bh = row / p.Wd;
    const int ho = (int)(bh % p.H_out), b = (int)(bh / p.H_out);
    float o[8];
    const int col0 = g * 8;
    if ((p.C & 7) == 0 && col0 < kc) {                       // the 8 columns lie inside one kh block: one 16-byte (32-byte) load
      const int kh = col0 / p.C, c = col0 - kh * p.C;
      bool valid;
      const int hi = pad_index(p.stride * ho + kh - p.pad, p.H_in, p.pad_mode, valid);
      if (valid) {
        const int64_t src = (((int64_t)b * p.H_in + hi) * p.Wd + w) * p.ldx + c;
        if (p.x_f32) load8<float>((const float*)p.x + src, o);
        else load8<T>((const T*)p.x + src, o);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = col0 + e;
        float v = 0.f;
        if (col < kc) {
          const int kh = col / p.C, c = col - kh * p.C;
          bool valid;
          const int hi = pad_index(p.stride * ho + kh - p.pad, p.H_in, p.pad_mode, valid);
          if (valid) {
            const int64_t src = (((int64_t)b * p.H_in + hi) * p.Wd + w) * p.ldx + c;
            v = p.x_f32 ? ((const float*)p.x)[src] : Elem<T>::ld((const T*)p.x + src);
          }
        }
        o[e] = v;
      }
    }
    store8<T>((T*)p.out + row * p.ldo + col0, o);
  }
}

// im2col along both axes (ZsConv2dGather.full): one thread per (output row (b, ho, wo), 8-column group)
template <typename T>
__global__ __launch_bounds__(NTP) void conv2d_im2col_kernel(const ZsConv2dGather p, int W_out) {
  const int groups = (int)(p.ldo / 8);
  const int64_t total = (int64_t)p.B * p.H_out * W_out * groups;
  const int kkc = p.k * p.k * p.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const int64_t row = i / groups;
    const int wo = (int)(row % W_out);
    const int64_t bh = row / W_out;
    const int ho = (int)(bh % p.H_out), b = (int)(bh / p.H_out);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = g * 8 + e;
      float v = 0.f;
      if (col < kkc) {
        const int tap = col / p.C, c = col - tap * p.C;
        const int kh = tap / p.k, kw = tap - kh * p.k;
        bool vh, vw;
        const int hi = pad_index(p.stride * ho + kh - p.pad, p.H_in, p.pad_mode, vh);
        const int wi = pad_index(p.stride * wo + kw - p.pad, p.Wd, p.pad_mode, vw);
        if (vh && vw) {
          const int64_t src = (((int64_t)b * p.H_in + hi) * p.Wd + wi) * p.ldx + c;
          v = p.x_f32 ? ((const float*)p.x)[src] : Elem<T>::ld((const T*)p.x + src);
        }
      }
      o[e] = v;
    }
    store8<T>((T*)p.out + row * p.ldo + g * 8, o);
  }
}

// The first layer without the im2col buffer (zs_conv1_fwd): a wave = 16 consecutive output positions x all (<= 64) output channels.
// MFMA operands: A = weight rows (physical row i of channel tile t is channel 16 (i / 4) + 4 t + i % 4, so that a lane ends up with
// 16 CONSECUTIVE channels of its position), B = the im2col column of the lane's position, 8 taps per lane gathered from the image.
constexpr int C1F_TPW = 8;                                                          // 16-position tiles per wave (weights loaded once)
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const ZsConv1Fwd p, int Ho, int Wo, int64_t M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  // weight fragments and bias of this lane's channels
  uint4 wfr[4];
  float bs[16];
  const bf16_t* Wb = (const bf16_t*)p.W;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ch = 16 * (j >> 2) + 4 * t + (j & 3);                              // A-operand row j of tile t
    wfr[t] = ch < p.Cout ? *reinterpret_cast<const uint4*>(Wb + (int64_t)ch * p.ldw + 8 * q) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) bs[e] = (p.bias != nullptr && 16 * q + e < p.Cout) ? p.bias[16 * q + e] : 0.f;
  const int pad = p.k >> 1, kk = p.k * p.k;
  int khs[8], kws[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int tap = 8 * q + e; khs[e] = tap / p.k; kws[e] = tap - khs[e] * p.k; }
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * C1F_TPW;
#pragma unroll 2
  for (int it = 0; it < C1F_TPW; ++it) {
    const int64_t m = (tile0 + it) * 16 + j;
    if ((tile0 + it) * 16 >= M) break;                                           // (wave-uniform)
    // the lane's position and its 8 taps k = 8 q + e
    const int mc = (int)(m < M ? m : M - 1);                                     // (M < 2^31: host check; 32-bit divisions)
    const int bh = mc / Wo, wo = mc - bh * Wo;
    const int b = bh / Ho, ho = bh - b * Ho;
    const float* xb = p.x + (int64_t)b * p.H * p.Wd;
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bool vh, vw;
      const int hi = pad_index(2 * ho + khs[e] - pad, p.H, p.pad_mode, vh);
      const int wi = pad_index(2 * wo + kws[e] - pad, p.Wd, p.pad_mode, vw);
      xv[e] = (8 * q + e < kk && vh && vw) ? xb[(int64_t)hi * p.Wd + wi] : 0.f;
    }
    uint4 bfr;
    bfr.x = (uint32_t)f2bf(xv[0]) | ((uint32_t)f2bf(xv[1]) << 16);
    bfr.y = (uint32_t)f2bf(xv[2]) | ((uint32_t)f2bf(xv[3]) << 16);
    bfr.z = (uint32_t)f2bf(xv[4]) | ((uint32_t)f2bf(xv[5]) << 16);
    bfr.w = (uint32_t)f2bf(xv[6]) | ((uint32_t)f2bf(xv[7]) << 16);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[t]), __builtin_bit_cast(bf16x8, bfr), acc[t], 0, 0, 0);
    }
    if (m >= M || 16 * q >= p.Cout) continue;
    // acc[t][r] = channel 16 q + 4 t + r of position j
    float lo[8], hi[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[t][r] + bs[4 * t + r];
        if (p.act == ZS_ACT_LRELU) v = lrelu_f(v, p.slope);
        if (t < 2) lo[4 * t + r] = v; else hi[4 * (t - 2) + r] = v;
      }
    bf16_t* dst = (bf16_t*)p.out + m * p.ldo + 16 * q;
    store8<bf16_t>(dst, lo);
    store8<bf16_t>(dst + 8, hi);
  }
}

// Weight gradient of the first layer without the im2col buffer (zs_conv1_wgrad).  K of the MFMA = positions: a wave takes tiles of
// 32 positions (lane (j, q): positions 8 q .. 8 q + 7 of the tile); A = gz transposed (row = channel 16 ct + j), B = the im2col
// columns (column = tap 16 kt + j; tap k*k is a column of ones: its result row is the bias gradient).
constexpr int C1W_BLOCKS = 1024;
constexpr int C1W_PITCH = 68;                                                       // bf16 elements per staged gz row (136 B: the four
                                                                                    // 8-row groups of a fragment read land on banks 16 apart)
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const ZsConv1Wgrad p, int Ho, int Wo, int64_t M) {
  __shared__ float red[4][64 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int pad = p.k >> 1, kk = p.k * p.k;
  // the wave's staging tile of gz ([32 positions][64 channels] bf16) lives in its slice of `red` until the final reduction
  bf16_t* st = reinterpret_cast<bf16_t*>(&red[wave][0]);
  f32x4 acc[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) acc[ct][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int kh[2], kw[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) { const int tap = 16 * kt + j; kh[kt] = tap / p.k; kw[kt] = tap - kh[kt] * p.k; }
  const bf16_t* gz = (const bf16_t*)p.gz;
  const bool vec = (p.Cout == 64) && ((p.ldg & 7) == 0) && ((((uintptr_t)p.gz) & 15) == 0);
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t per_wave = (n_tiles + (int64_t)C1W_BLOCKS * 4 - 1) / ((int64_t)C1W_BLOCKS * 4);
  const int64_t t_lo = ((int64_t)blockIdx.x * 4 + wave) * per_wave, t_hi = min(n_tiles, t_lo + per_wave);      // contiguous, fixed
  for (int64_t tile = t_lo; tile < t_hi; ++tile) {
    const int64_t mt = tile * 32;
    // stage the tile: lane l loads 16-byte pieces (row l / 8 + 8 i, channels 8 (l % 8) ..): coalesced 128-byte rows
    if (vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (lane >> 3) + 8 * i, c8 = (lane & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (mt + r < M) v = *reinterpret_cast<const uint4*>(gz + (mt + r) * p.ldg + c8);
        *reinterpret_cast<uint2*>(st + r * C1W_PITCH + c8) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(st + r * C1W_PITCH + c8 + 4) = make_uint2(v.z, v.w);
      }
    } else {
      for (int i = lane; i < 32 * 64; i += 64) {
        const int r = i >> 6, cc = i & 63;
        st[r * C1W_PITCH + cc] = (mt + r < M && cc < p.Cout) ? gz[(mt + r) * p.ldg + cc] : (bf16_t)0;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int64_t m0 = mt + 8 * q;
    const int mm = (int)(m0 < M ? m0 : M - 1);                                  // (M < 2^31: host check; 32-bit divisions)
    int bh = mm / Wo, wo = mm - bh * Wo;
    int b = bh / Ho, ho = bh - b * Ho;
    float bv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = m0 + e < M;
      const float* xb = p.x + (int64_t)b * p.H * p.Wd;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int tap = 16 * kt + j;
        float v = 0.f;
        if (in && tap < kk) {
          bool vh, vw;
          const int hi = pad_index(2 * ho + kh[kt] - pad, p.H, p.pad_mode, vh);
          const int wi = pad_index(2 * wo + kw[kt] - pad, p.Wd, p.pad_mode, vw);
          if (vh && vw) v = xb[(int64_t)hi * p.Wd + wi];
        } else if (in && tap == kk) v = 1.f;                                   // the bias column
        bv[kt][e] = v;
      }
      if (++wo == Wo) { wo = 0; if (++ho == Ho) { ho = 0; ++b; } }
    }
    uint4 af[4], bf_[2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      uint32_t w4[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const uint32_t a0 = st[(8 * q + 2 * e2) * C1W_PITCH + 16 * ct + j], a1 = st[(8 * q + 2 * e2 + 1) * C1W_PITCH + 16 * ct + j];
        w4[e2] = a0 | (a1 << 16);
      }
      af[ct] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf_[kt].x = (uint32_t)f2bf(bv[kt][0]) | ((uint32_t)f2bf(bv[kt][1]) << 16); bf_[kt].y = (uint32_t)f2bf(bv[kt][2]) | ((uint32_t)f2bf(bv[kt][3]) << 16);
      bf_[kt].z = (uint32_t)f2bf(bv[kt][4]) | ((uint32_t)f2bf(bv[kt][5]) << 16); bf_[kt].w = (uint32_t)f2bf(bv[kt][6]) | ((uint32_t)f2bf(bv[kt][7]) << 16);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        acc[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ct]), __builtin_bit_cast(bf16x8, bf_[kt]), acc[ct][kt], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();                                           // the next tile's staging overwrites what was just read
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // acc[ct][kt][r] = (channel 16 ct + 4 q + r, tap 16 kt + j): waves summed in wave order, one partial per workgroup
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][(16 * ct + 4 * q + r) * 32 + 16 * kt + j] = acc[ct][kt][r];
  __syncthreads();
  float* out = p.workspace + (int64_t)blockIdx.x * (64 * 32);
  for (int i = threadIdx.x; i < 64 * 32; i += 256) out[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

// 64 workgroups x 32 outputs: thread (output o, eighth g) sums 128 partials in order, the eighths are combined in order
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(const ZsConv1Wgrad p) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;                                           // (channel, tap slot)
  float s = 0.f;
  for (int w = g * (C1W_BLOCKS / 8); w < (g + 1) * (C1W_BLOCKS / 8); ++w) s += p.workspace[(int64_t)w * (64 * 32) + i];
  part[g][o] = s;
  __syncthreads();
  if (g != 0) return;
  s = part[0][o];
#pragma unroll
  for (int k = 1; k < 8; ++k) s += part[k][o];
  const int co = i >> 5, tap = i & 31, kk = p.k * p.k;
  if (co >= p.Cout || tap > kk) return;
  if (tap < kk) {
    float* d = p.dW + (int64_t)co * p.lddw + tap;
    *d = p.accumulate ? *d + s : s;
  } else if (p.db != nullptr) {
    p.db[co] = p.accumulate ? p.db[co] + s : s;
  }
}

// transpose of the above (ZsConv2dFold.full): dX[b, hi, w, c] = sum over (ph in {hi, reflection partners}, kh -> ho) x
// (pw in {w, reflection partners}, kw -> wo) of gp[(b, ho, wo), (kh*k + kw)*C + c]; one thread per (b, hi, w, c), fixed order
template <typename T>
__global__ __launch_bounds__(NTP) void conv2d_col2im_kernel(const ZsConv2dFold p, int W_out) {
  const int64_t total = (int64_t)p.B * p.H_in * p.Wd * p.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % p.C);
    const int64_t row = i / p.C;
    const int w = (int)(row % p.Wd);
    const int64_t bh = row / p.Wd;
    const int hi = (int)(bh % p.H_in), b = (int)(bh / p.H_in);
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = hi;
    ws[nw++] = w;
    if (p.pad_mode == ZS_PAD_REFLECT) {
      if (hi >= 1 && hi <= p.pad) hs[nh++] = -hi;
      if (hi <= p.H_in - 2 && hi >= p.H_in - 1 - p.pad) hs[nh++] = 2 * (p.H_in - 1) - hi;
      if (w >= 1 && w <= p.pad) ws[nw++] = -w;
      if (w <= p.Wd - 2 && w >= p.Wd - 1 - p.pad) ws[nw++] = 2 * (p.Wd - 1) - w;
    }
    float acc = 0.f;
    for (int a = 0; a < nh; ++a)
      for (int kh = 0; kh < p.k; ++kh) {
        const int nh_ = hs[a] + p.pad - kh;
        if (nh_ < 0 || nh_ % p.stride != 0) continue;
        const int ho = nh_ / p.stride;
        if (ho >= p.H_out) continue;
        for (int q = 0; q < nw; ++q)
          for (int kw = 0; kw < p.k; ++kw) {
            const int nw_ = ws[q] + p.pad - kw;
            if (nw_ < 0 || nw_ % p.stride != 0) continue;
            const int wo = nw_ / p.stride;
            if (wo >= W_out) continue;
            acc += Elem<T>::ld((const T*)p.gp + (((int64_t)b * p.H_out + ho) * W_out + wo) * p.ldg + (int64_t)(kh * p.k + kw) * p.C + c);
          }
      }
    if (p.add) acc += Elem<T>::ld((const T*)p.add + row * p.ldadd + c);
    if (p.out_f32) ((float*)p.out)[row * p.ldo + c] = acc;
    else Elem<T>::st((T*)p.out + row * p.ldo + c, acc);
  }
}

// ---- Conv2d data gradient: fold the W-padded / H-gathered gradient back -----------------------------------------------------
// dX[b, hi, w, c] = sum over (ph in {hi and its reflection partners}, kh with (ph + pad - kh) % stride == 0 -> ho) and over
//                   (wp in {w + pad and its reflection partners}) of gp[(b, ho), wp, kh*C + c]
// grid (column blocks of (w, 8-channel group), image rows (b, hi)): the row decomposition is per workgroup, and for an interior
// position (no reflection partner in either axis) the contributing taps kh = r, r + stride, ... and their rows ho are the same for
// the whole workgroup: their loads are issued together and summed in ascending kh (the order of the general loop, which the
// positions next to the padding still take).  As a flat loop with 64-bit div / mod per element and one dependent load after the
// other the layer-2 fold ran at 2.7 TB/s.
template <typename T>
__global__ __launch_bounds__(NTP) void conv2d_fold_kernel(const ZsConv2dFold p) {
  const int cols = p.out_f32 ? p.C : p.fill_cols;
  const int groups = (cols + 7) / 8;
  const int Wp = p.gp_rows > 0 ? p.gp_rows : p.Wd + 2 * p.pad;
  const int t = blockIdx.x * NTP + threadIdx.x;
  if (t >= p.Wd * groups) return;
  const int w = t / groups, g = t - w * groups;
  const int bh = blockIdx.y;
  const int b = bh / p.H_in, hi = bh - b * p.H_in;
  const int64_t row = (int64_t)bh * p.Wd + w;
  const int c0 = g * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // candidate positions in the (unpadded-coordinate) extended domains
  int hs[3], ws[3], nh = 0, nw = 0;
  hs[nh++] = hi;
  ws[nw++] = w;
  if (p.pad_mode == ZS_PAD_REFLECT) {
    if (hi >= 1 && hi <= p.pad) hs[nh++] = -hi;
    if (hi <= p.H_in - 2 && hi >= p.H_in - 1 - p.pad) hs[nh++] = 2 * (p.H_in - 1) - hi;
    if (w >= 1 && w <= p.pad) ws[nw++] = -w;
    if (w <= p.Wd - 2 && w >= p.Wd - 1 - p.pad) ws[nw++] = 2 * (p.Wd - 1) - w;
  }
  const int r0 = (hi + p.pad) % p.stride, ho0 = (hi + p.pad - r0) / p.stride;       // taps kh = r0 + stride j  <->  rows ho0 - j
  if (nh == 1 && nw == 1 && (p.C & 7) == 0 && r0 + 4 * p.stride >= p.k) {           // interior, at most 4 taps
    const T* base = (const T*)p.gp + (int64_t)(w + p.pad) * p.ldg + c0;
    float v[4][8];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kh = r0 + p.stride * j, ho = ho0 - j;
      ok[j] = kh < p.k && ho >= 0 && ho < p.H_out;                                   // (uniform over the workgroup)
      if (ok[j]) load8<T>(base + ((int64_t)b * p.H_out + ho) * Wp * p.ldg + (int64_t)kh * p.C, v[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ok[j]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[j][e];
      }
  } else {
    for (int a = 0; a < nh; ++a) {
      for (int kh = 0; kh < p.k; ++kh) {
        const int num = hs[a] + p.pad - kh;
        if (num < 0 || num % p.stride != 0) continue;
        const int ho = num / p.stride;
        if (ho >= p.H_out) continue;
        for (int q = 0; q < nw; ++q) {
          const int wp = ws[q] + p.pad;                       // index in the padded W domain
          const T* src = (const T*)p.gp + (((int64_t)b * p.H_out + ho) * Wp + wp) * p.ldg + (int64_t)kh * p.C + c0;
          if ((p.C & 7) == 0) {
            float v[8];
            load8<T>(src, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c0 + e < p.C) acc[e] += Elem<T>::ld(src + e);
          }
        }
      }
    }
  }
  if (p.add) {
    const T* ad = (const T*)p.add + row * p.ldadd + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < p.C) acc[e] += Elem<T>::ld(ad + e);
  }
  if (p.out_f32) {
    float* o = (float*)p.out + row * p.ldo + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < p.C) o[e] = acc[e];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e >= p.C) acc[e] = 0.f;
    store8<T>((T*)p.out + row * p.ldo + c0, acc);
  }
}

// ---- un-padding of a parity-class data gradient (zs_conv2d_unpad) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTP) void conv2d_unpad_kernel(const ZsConv2dUnpad p) {
  const int groups = (p.fill_cols + 7) / 8;
  const int64_t total = (int64_t)p.B * p.H * p.W * groups;
  const int Hc0 = (p.Hp + 1) >> 1, Hc1 = p.Hp >> 1, Wc0 = (p.Wp + 1) >> 1, Wc1 = p.Wp >> 1;
  const void* const gq[4] = {p.g00, p.g01, p.g10, p.g11};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const int64_t row = i / groups;
    const int w = (int)(row % p.W);
    const int64_t bh = row / p.W;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const int c0 = g * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < p.C) {
      int hs[3], ws[3], nh = 0, nw = 0;
      hs[nh++] = h;
      ws[nw++] = w;
      if (p.pad_mode == ZS_PAD_REFLECT) {
        if (h >= 1 && h <= p.pad) hs[nh++] = -h;
        if (h <= p.H - 2 && h >= p.H - 1 - p.pad) hs[nh++] = 2 * (p.H - 1) - h;
        if (w >= 1 && w <= p.pad) ws[nw++] = -w;
        if (w <= p.W - 2 && w >= p.W - 1 - p.pad) ws[nw++] = 2 * (p.W - 1) - w;
      }
      for (int a = 0; a < nh; ++a) {
        const int hp = hs[a] + p.pad;
        if (hp < 0 || hp >= p.Hp) continue;
        const int ph = hp & 1, Hc = ph ? Hc1 : Hc0;
        for (int q = 0; q < nw; ++q) {
          const int wp = ws[q] + p.pad;
          if (wp < 0 || wp >= p.Wp) continue;
          const int pw = wp & 1, Wc = pw ? Wc1 : Wc0;
          const T* src = (const T*)gq[2 * ph + pw] + (((int64_t)b * Hc + (hp >> 1)) * Wc + (wp >> 1)) * p.ldg + c0;
          float v[8];
          load8<T>(src, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      }
      if (p.add) {
        float v[8];
        load8<T>((const T*)p.add + row * p.ldadd + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e >= p.C) acc[e] = 0.f;
    }
    store8<T>((T*)p.out + row * p.ldo + c0, acc);
  }
}

// ---- per-(b, c) moments over T rows --------------------------------------------------------------------------------------------
// grid (C/64 chunks, B, slabs of MOM_ROWS rows); thread (cg = tid & 7, rg = tid >> 3): 8 channels of rows rg, rg + 32, ...
template <typename T>
__global__ __launch_bounds__(NTP) void row_moments_kernel(const ZsRowMoments p, int nslab) {
  __shared__ float red[3][32 * 64];
  const int tid = threadIdx.x, cg = tid & 7, rg = tid >> 3;
  const int b = blockIdx.y, slab = blockIdx.z, c0 = blockIdx.x * 64 + cg * 8;
  const bool cvalid = c0 < p.C;
  float s1[8], s2[8], s3[8], ctr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; s3[e] = 0.f; ctr[e] = 0.f; }
  if (cvalid && p.center_sum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ctr[e] = (c0 + e < p.C) ? p.center_sum[(int64_t)b * p.C + c0 + e] * p.center_scale : 0.f;
  }
  const int t_lo = slab * MOM_ROWS, t_hi = min(p.T, t_lo + MOM_ROWS);
  if (cvalid) {
    for (int t = t_lo + rg; t < t_hi; t += 32) {
      const int64_t row = (int64_t)b * p.T + t;
      float u[8], ur[8];
      load8<T>((const T*)p.u + row * p.ldu + c0, ur);
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = ur[e] - ctr[e];
      if (p.y) {
        float yv[8];
        load8<T>((const T*)p.y + row * p.ldy + c0, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] *= dlrelu_f(yv[e], p.slope);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] += u[e];
      if (p.s2) {
        if (p.v == p.u) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s2[e] += u[e] * u[e];
        } else {
          float v[8];
          load8<T>((const T*)p.v + row * p.ldv + c0, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) s2[e] += u[e] * v[e];
        }
      }
      if (p.w) {
        float wv[8];
        load8<T>((const T*)p.w + row * p.ldw + c0, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s3[e] += ur[e] * wv[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][rg * 64 + cg * 8 + e] = s1[e]; red[1][rg * 64 + cg * 8 + e] = s2[e]; red[2][rg * 64 + cg * 8 + e] = s3[e]; }
  __syncthreads();
  if (tid < 192) {
    const int m = tid >> 6, c = tid & 63;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) s += red[m][r * 64 + c];
    const int cc = blockIdx.x * 64 + c;
    if (cc < p.C) p.partial[(((int64_t)b * nslab + slab) * 3 + m) * p.C + cc] = s;
  }
}

__global__ void row_moments_finish_kernel(const ZsRowMoments p, int nslab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)p.B * p.C) return;
  const int b = (int)(i / p.C), c = (int)(i % p.C);
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int s = 0; s < nslab; ++s) {
    const float* q = p.partial + ((int64_t)b * nslab + s) * 3 * p.C + c;
    a1 += q[0]; a2 += q[p.C]; a3 += q[2 * (int64_t)p.C];
  }
  if (p.s1) p.s1[i] = a1;
  if (p.s2) p.s2[i] = a2;
  if (p.s3) p.s3[i] = a3;
}

// ---- InstanceNorm2d forward statistics in one pass (zs_in2d_stats) ----------------------------------------------------------
// grid (C/64, B, slabs of MOM_ROWS rows); thread (cg, rg): 8 channels of the slab's rows rg, rg + 32, ... (<= 16 rows, kept in
// registers): thread-local mean, then thread-local centred second moment, merged over the 32 row groups in a fixed order.
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
  if (nb <= 0.f) return;
  const float nt = n + nb, d = meanb - mean;
  mean += d * (nb / nt);
  m2 += m2b + d * d * (n * nb / nt);
  n = nt;
}

template <typename T>
__global__ __launch_bounds__(NTP) void in2d_stats_kernel(const void* y_, int64_t ldy, int T_, int C, float* partial, int nslab) {
  constexpr int RP = MOM_ROWS / 32;
  __shared__ float red[2][32 * 64];
  __shared__ float cnt[32];
  const int tid = threadIdx.x, cg = tid & 7, rg = tid >> 3;
  const int b = blockIdx.y, slab = blockIdx.z, c0 = blockIdx.x * 64 + cg * 8;
  const bool cvalid = c0 < C;
  const int t_lo = slab * MOM_ROWS, t_hi = min(T_, t_lo + MOM_ROWS);
  Raw8<T> v[RP];
  int nrow = 0;
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const int t = t_lo + rg + 32 * i;
    v[i].zero();
    if (cvalid && t < t_hi) { v[i].ld((const T*)y_ + ((int64_t)b * T_ + t) * ldy + c0); ++nrow; }
  }
  float mean[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { mean[e] = 0.f; m2[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    float f[8]; v[i].cvt(f);                                   // rows past the slab are zeros
#pragma unroll
    for (int e = 0; e < 8; ++e) mean[e] += f[e];
  }
  const float inv = nrow > 0 ? 1.f / (float)nrow : 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] *= inv;
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    if (t_lo + rg + 32 * i < t_hi) {
      float f[8]; v[i].cvt(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean[e]; m2[e] += d * d; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][rg * 64 + cg * 8 + e] = mean[e]; red[1][rg * 64 + cg * 8 + e] = m2[e]; }
  if (cg == 0) cnt[rg] = (float)nrow;                          // (the same for every channel group of a row group)
  __syncthreads();
  if (tid < 64) {
    float n = 0.f, mu = 0.f, q = 0.f;
    for (int r = 0; r < 32; ++r) {
      const float nb = cnt[r];
      if (n == 0.f) { n = nb; mu = red[0][r * 64 + tid]; q = red[1][r * 64 + tid]; }
      else chan_merge(n, mu, q, nb, red[0][r * 64 + tid], red[1][r * 64 + tid]);
    }
    const int cc = blockIdx.x * 64 + tid;
    if (cc < C) {
      float* o = partial + (((int64_t)b * nslab + slab) * 3) * C + cc;
      o[0] = n; o[C] = mu; o[2 * (int64_t)C] = q;
    }
  }
}

__global__ void in2d_stats_finish_kernel(const float* partial, int B, int C, int nslab, float eps, float* mean, float* rstd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * C) return;
  const int b = (int)(i / C), c = (int)(i % C);
  float n = 0.f, mu = 0.f, q = 0.f;
  for (int s = 0; s < nslab; ++s) {
    const float* o = partial + ((int64_t)b * nslab + s) * 3 * C + c;
    if (n == 0.f) { n = o[0]; mu = o[C]; q = o[2 * (int64_t)C]; }
    else chan_merge(n, mu, q, o[0], o[C], o[2 * (int64_t)C]);
  }
  mean[i] = mu;
  rstd[i] = 1.0f / sqrtf(q / n + eps);
}

__global__ void in2d_finalize_kernel(const float* s1, const float* q, float* mean, float* rstd, int64_t n, float invT, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mean[i] = s1[i] * invT;
  rstd[i] = 1.0f / sqrtf(q[i] * invT + eps);
}

// ---- row-streaming elementwise kernels ---------------------------------------------------------------------------------------------
// grid (row chunks, sample); a thread keeps ONE group of 8 channels of ONE sample for all its rows, so the per-(sample, channel)
// statistics are loaded once per thread -- as a flat loop over (row, group) every 16-byte payload access dragged 24-40 scalar
// look-ups of mean / rstd / moments behind it and the kernels ran at 1.3-1.4 TB/s on the first layer's 269 MB planes.
struct RowWalk {
  int g, c0, b, r, step, ok;
  __device__ __forceinline__ RowWalk(int C) {
    const int groups = (C + 7) / 8, rpp = NTP / groups;                       // groups <= NTP (host check)
    g = threadIdx.x % groups; c0 = g * 8; b = blockIdx.y;
    const int r_in = threadIdx.x / groups;
    ok = r_in < rpp;
    r = blockIdx.x * rpp + r_in; step = gridDim.x * rpp;
  }
};

template <typename T>
__global__ __launch_bounds__(NTP) void in2d_fwd_kernel(const ZsIn2dFwd p) {
  const RowWalk w(p.C);
  if (!w.ok) return;
  float mu[8], rs[8], dm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = w.c0 + e;
    const int64_t bc = (int64_t)w.b * p.C + c;
    mu[e] = c < p.C ? p.mean[bc] : 0.f; rs[e] = c < p.C ? p.rstd[bc] : 0.f; dm[e] = (c < p.C) ? (p.dm ? p.dm[bc] : 1.f) : 0.f;
  }
  for (int t = w.r; t < p.T; t += w.step) {
    const int64_t row = (int64_t)w.b * p.T + t;
    float y[8], o[8];
    load8<T>((const T*)p.y + row * p.ldy + w.c0, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (y[e] - mu[e]) * rs[e] * dm[e];
    store8<T>((T*)p.a + row * p.lda + w.c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(NTP) void in2d_bwd_kernel(const ZsIn2dBwd p) {
  const RowWalk w(p.C);
  if (!w.ok) return;
  const float invT = 1.0f / (float)p.T;
  float rd[8], s1t[8], r[8], inv_dm[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = w.c0 + e;
    const bool in = c < p.C;
    const int64_t bc = in ? (int64_t)w.b * p.C + c : 0;
    const float dm = p.dm ? p.dm[bc] : 1.f;
    r[e] = in ? p.rstd[bc] : 0.f;
    inv_dm[e] = dm > 0.f ? 1.f / dm : 0.f;
    s2[e] = p.S2[bc] + (p.S2x ? p.S2x[bc] : 0.f);
    rd[e] = r[e] * dm;
    s1t[e] = p.S1[bc] * invT;
  }
  for (int t = w.r; t < p.T; t += w.step) {
    const int64_t row = (int64_t)w.b * p.T + t;
    float ga[8], a[8], y[8], o[8];
    load8<T>((const T*)p.ga + row * p.ldga + w.c0, ga);
    if (p.ga2) {
      float g2[8];
      load8<T>((const T*)p.ga2 + row * p.ldga2 + w.c0, g2);
#pragma unroll
      for (int e = 0; e < 8; ++e) ga[e] += g2[e];
    }
    load8<T>((const T*)p.a + row * p.lda + w.c0, a);
    load8<T>((const T*)p.y + row * p.ldy + w.c0, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gy = rd[e] * (ga[e] - s1t[e]) - r[e] * a[e] * inv_dm[e] * s2[e] * invT;
      o[e] = (w.c0 + e < p.C) ? gy * dlrelu_f(y[e], p.slope) : 0.f;
    }
    store8<T>((T*)p.gz + row * p.ldgz + w.c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(NTP) void in2d_adj_kernel(const ZsIn2dAdj p) {
  const RowWalk w(p.C);
  if (!w.ok) return;
  const float invT = 1.0f / (float)p.T;
  float dmr[8], nir[8], inv_dm[8], a1t[8], m3[8], s2[8], dmv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = w.c0 + e;
    const bool in = c < p.C;
    const int64_t bc = in ? (int64_t)w.b * p.C + c : 0;
    const float dm = p.dm ? p.dm[bc] : 1.f, r = in ? p.rstd[bc] : 0.f;
    dmv[e] = dm;
    inv_dm[e] = dm > 0.f ? 1.f / dm : 0.f;
    m3[e] = inv_dm[e] * p.A2[bc] * invT;                                      // mean(gbar_y * xhat)
    a1t[e] = p.A1[bc] * invT;
    s2[e] = p.S2[bc];
    dmr[e] = dm * r;
    nir[e] = -inv_dm[e] * r;
  }
  for (int t = w.r; t < p.T; t += w.step) {
    const int64_t row = (int64_t)w.b * p.T + t;
    float gbz[8], y[8], a[8], ga[8], o1[8], o2[8];
    load8<T>((const T*)p.gbz + row * p.ldgbz + w.c0, gbz);
    load8<T>((const T*)p.y + row * p.ldy + w.c0, y);
    load8<T>((const T*)p.a + row * p.lda + w.c0, a);
    load8<T>((const T*)p.ga + row * p.ldga + w.c0, ga);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gby = gbz[e] * dlrelu_f(y[e], p.slope);
      const float xhat = a[e] * inv_dm[e];
      const bool in = w.c0 + e < p.C;
      o1[e] = in ? dmr[e] * (gby - a1t[e] - xhat * m3[e]) : 0.f;
      o2[e] = in ? nir[e] * (gby * s2[e] * invT + ga[e] * dmv[e] * m3[e]) : 0.f;
    }
    store8<T>((T*)p.gba + row * p.ldgba + w.c0, o1);
    store8<T>((T*)p.xba + row * p.ldxba + w.c0, o2);
  }
}

// ---- gradient penalty glue ----------------------------------------------------------------------------------------------------------
__global__ void lerp_rows_kernel(const float* x, const float* y, const float* alpha, float* out, int B, int64_t n) {
  const int64_t total = (int64_t)B * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = alpha[i / n];
    out[i] = a * x[i] + (1.f - a) * y[i];
  }
}

// one 1024-thread workgroup per sample: s_b = sqrt(1e-12 + sum g^2) in double, fixed order
__global__ __launch_bounds__(1024) void gp_norm_kernel(const float* g, int64_t n, float* s_out) {
  __shared__ double red[16];
  const float* gb = g + (int64_t)blockIdx.x * n;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) { const double v = (double)gb[i]; acc += v * v; }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    s_out[blockIdx.x] = (float)sqrt(1e-12 + t);
  }
}

__global__ void gp_finish_kernel(const float* g, const float* s, int B, int64_t n, float scale, float* gp_out, float* gbar) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && gp_out) {
    float t = 0.f;
    for (int b = 0; b < B; ++b) { const float d = 1.f - s[b]; t += d * d; }
    *gp_out = t / (float)B;
  }
  if (!gbar) return;
  const int64_t total = (int64_t)B * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float sb = s[i / n];
    gbar[i] = scale * (2.f / (float)B) * (sb - 1.f) / sb * g[i];
  }
}

__global__ void gen_combine_fwd_kernel(const float* xd, const float* m, int64_t ld_in, float* xg, int64_t rows, int F, int mode) {
  const int64_t n = rows * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = (i / F) * ld_in + (i % F);
    xg[i] = mode ? xd[k] + xd[k] * m[k] : xd[k] + m[k];
  }
}

template <typename T>
__global__ void gen_combine_bwd_kernel(const ZsGenCombineBwd p) {
  const int64_t total = p.rows * p.fill_cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / p.fill_cols;
    const int c = (int)(i % p.fill_cols);
    float v = 0.f;
    if (c < p.F) {
      const int64_t k = r * p.ld_in + c;
      const float m = p.m[k];
      v = p.dx_gen[r * p.ld_dx + c] * (p.mode ? p.xd[k] : 1.f) * (p.tanh_out ? (1.f - m * m) : m * (1.f - m));
    }
    Elem<T>::st((T*)p.dpre + r * p.ldo + c, v);
  }
}

__global__ __launch_bounds__(NTP) void l1_plain_stage1(const float* a, const float* b, int64_t n, float scale, float* partial, float* d) {
  __shared__ float red[NTP / 64];
  float s = 0.f;
  const float gs = scale / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float df = a[i] - b[i];
    s += fabsf(df);
    if (d) d[i] = df > 0.f ? gs : (df < 0.f ? -gs : 0.f);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void l1_plain_stage2(const float* partial, int nb, int64_t n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nb; ++i) t += (double)partial[i];
    *out = (float)(t / (double)n);
  }
}

// grid of the per-sample row walkers (RowWalk): enough row chunks per sample that the launch has a few thousand workgroups and
// every thread still keeps its channel constants for several rows
dim3 row_walk_grid(int B, int T, int C) {
  const int groups = (C + 7) / 8, rpp = NTP / groups;
  int64_t passes = ((int64_t)T + rpp - 1) / rpp;
  int64_t nbx = (passes + 7) / 8;                                            // ~8 rows per thread
  const int64_t want = (4096 + B - 1) / B;                                  // ... but at least ~4096 workgroups in the launch
  if (nbx < want) nbx = want;
  if (nbx > passes) nbx = passes;
  if (nbx < 1) nbx = 1;
  if (nbx > 65535) nbx = 65535;
  return dim3((unsigned)nbx, (unsigned)B);
}

unsigned grid_for(int64_t total) {
  int64_t nb = (total + NTP - 1) / NTP;
  if (nb > 16384) nb = 16384;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}
bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

#define ZS_DT_OK(p) ZS_REQUIRE((p)->dtype == ZS_F32 || (p)->dtype == ZS_BF16, "bad dtype")

extern "C" int zs_conv2d_gather(const ZsConv2dGather* p, void* stream) {
  ZS_REQUIRE(p && p->x && p->out, "zs_conv2d_gather: null operand");
  ZS_DT_OK(p);
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE(p->B > 0 && p->H_in > 0 && p->H_out > 0 && p->Wd > 0 && p->C > 0 && p->k > 0 && p->stride > 0 && p->pad >= 0, "zs_conv2d_gather: sizes");
  if (p->full) {
    const int W_out = (p->Wd + 2 * p->pad - p->k) / p->stride + 1;
    ZS_REQUIRE(p->ldo >= (int64_t)p->k * p->k * p->C && p->ldo % 8 == 0 && al16p(p->out) && W_out > 0, "zs_conv2d_gather(full): ldo %lld must be a multiple of 8 >= k*k*C", (long long)p->ldo);
    ZS_REQUIRE(p->pad_mode != ZS_PAD_REFLECT || (p->pad < p->H_in && p->pad < p->Wd), "zs_conv2d_gather: Padding size should be less than the corresponding input dimension");
    ZS_REQUIRE((p->stride * (p->H_out - 1) + p->k - 1 - p->pad) < p->H_in + p->pad, "zs_conv2d_gather: H_out too large");
    const int64_t tot = (int64_t)p->B * p->H_out * W_out * (p->ldo / 8);
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(conv2d_im2col_kernel<float>, dim3(grid_for(tot)), dim3(NTP), 0, (hipStream_t)stream, *p, W_out);
    else hipLaunchKernelGGL(conv2d_im2col_kernel<bf16_t>, dim3(grid_for(tot)), dim3(NTP), 0, (hipStream_t)stream, *p, W_out);
    return zs_check_launch("zs_conv2d_gather.full");
  }
  ZS_REQUIRE(p->ldo >= (int64_t)p->k * p->C && p->ldo % 8 == 0 && al16p(p->out), "zs_conv2d_gather: ldo %lld must be a multiple of 8 >= k*C", (long long)p->ldo);
  ZS_REQUIRE(p->pad_mode != ZS_PAD_REFLECT || p->pad < p->H_in, "zs_conv2d_gather: Padding size should be less than the corresponding input dimension");
  ZS_REQUIRE((p->stride * (p->H_out - 1) + p->k - 1 - p->pad) < p->H_in + p->pad, "zs_conv2d_gather: H_out too large");
  if ((p->C & 7) == 0) ZS_REQUIRE(al16p(p->x) && (p->ldx * (p->x_f32 ? 4 : es)) % 16 == 0, "zs_conv2d_gather: alignment");
  const int64_t total = (int64_t)p->B * p->H_out * p->Wd * (p->ldo / 8);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(conv2d_gather_kernel<float>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(conv2d_gather_kernel<bf16_t>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_conv2d_gather");
}

extern "C" int zs_conv1_fwd(const ZsConv1Fwd* p, void* stream) {
  ZS_REQUIRE(p && p->x && p->W && p->out, "zs_conv1_fwd: null operand");
  ZS_REQUIRE(p->B > 0 && p->H > 0 && p->Wd > 0 && p->k > 0 && (p->k & 1) && p->k * p->k <= 32 && p->Cout > 0 && p->Cout % 16 == 0 && p->Cout <= 64,
             "zs_conv1_fwd: sizes (odd k with k*k <= 32, Cout a multiple of 16 <= 64)");
  ZS_REQUIRE(p->ldw >= 32 && p->ldw % 8 == 0 && al16p(p->W) && al16p(p->out) && p->ldo % 8 == 0 && p->ldo >= p->Cout, "zs_conv1_fwd: alignment / pitches");
  ZS_REQUIRE(p->act == ZS_ACT_NONE || p->act == ZS_ACT_LRELU, "zs_conv1_fwd: act");
  const int pad = p->k / 2;
  ZS_REQUIRE(p->pad_mode != ZS_PAD_REFLECT || (pad < p->H && pad < p->Wd), "zs_conv1_fwd: Padding size should be less than the corresponding input dimension");
  const int Ho = (p->H + 2 * pad - p->k) / 2 + 1, Wo = (p->Wd + 2 * pad - p->k) / 2 + 1;
  const int64_t M = (int64_t)p->B * Ho * Wo;
  const int64_t blocks = (M + 64 * C1F_TPW - 1) / (64 * C1F_TPW);
  ZS_REQUIRE(M < (1ll << 31) - 64 * C1F_TPW, "zs_conv1_fwd: too many output positions");
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p, Ho, Wo, M);
  return zs_check_launch("zs_conv1_fwd");
}

extern "C" size_t zs_conv1_wgrad_workspace(void) { return (size_t)C1W_BLOCKS * 64 * 32 * sizeof(float); }

extern "C" int zs_conv1_wgrad(const ZsConv1Wgrad* p, void* stream) {
  ZS_REQUIRE(p && p->x && p->gz && p->dW && p->workspace, "zs_conv1_wgrad: null operand");
  ZS_REQUIRE(p->B > 0 && p->H > 0 && p->Wd > 0 && p->k > 0 && (p->k & 1) && p->k * p->k < 32 && p->Cout > 0 && p->Cout % 16 == 0 && p->Cout <= 64,
             "zs_conv1_wgrad: sizes (odd k with k*k < 32, Cout a multiple of 16 <= 64)");
  ZS_REQUIRE(p->ldg >= p->Cout && p->lddw >= p->k * p->k && p->workspace_bytes >= zs_conv1_wgrad_workspace(), "zs_conv1_wgrad: pitches / workspace");
  const int pad = p->k / 2;
  ZS_REQUIRE(p->pad_mode != ZS_PAD_REFLECT || (pad < p->H && pad < p->Wd), "zs_conv1_wgrad: Padding size should be less than the corresponding input dimension");
  const int Ho = (p->H + 2 * pad - p->k) / 2 + 1, Wo = (p->Wd + 2 * pad - p->k) / 2 + 1;
  const int64_t M = (int64_t)p->B * Ho * Wo;
  ZS_REQUIRE(M < (1ll << 31) - 64, "zs_conv1_wgrad: too many output positions");
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(C1W_BLOCKS), dim3(256), 0, (hipStream_t)stream, *p, Ho, Wo, M);
  int rc = zs_check_launch("zs_conv1_wgrad");
  if (rc) return rc;
  hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_conv1_wgrad.reduce");
}

extern "C" int zs_conv2d_fold(const ZsConv2dFold* p, void* stream) {
  ZS_REQUIRE(p && p->gp && p->out, "zs_conv2d_fold: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->H_in > 0 && p->H_out > 0 && p->Wd > 0 && p->C > 0 && p->k > 0 && p->stride > 0 && p->pad >= 0, "zs_conv2d_fold: sizes");
  if (p->full) {
    const int W_out = (p->Wd + 2 * p->pad - p->k) / p->stride + 1;
    ZS_REQUIRE(p->ldg >= (int64_t)p->k * p->k * p->C && W_out > 0, "zs_conv2d_fold(full): ldg");
    ZS_REQUIRE(p->out_f32 || p->fill_cols == p->C, "zs_conv2d_fold(full): T-dtype output rows must be exactly C wide (no zero fill)");
    const int64_t tot = (int64_t)p->B * p->H_in * p->Wd * p->C;
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(conv2d_col2im_kernel<float>, dim3(grid_for(tot)), dim3(NTP), 0, (hipStream_t)stream, *p, W_out);
    else hipLaunchKernelGGL(conv2d_col2im_kernel<bf16_t>, dim3(grid_for(tot)), dim3(NTP), 0, (hipStream_t)stream, *p, W_out);
    return zs_check_launch("zs_conv2d_fold.full");
  }
  ZS_REQUIRE(p->ldg >= (int64_t)p->k * p->C && al16p(p->gp), "zs_conv2d_fold: ldg");
  if (!p->out_f32) ZS_REQUIRE(p->fill_cols >= p->C && p->fill_cols % 8 == 0 && p->fill_cols <= p->ldo && al16p(p->out) && p->ldo % 8 == 0, "zs_conv2d_fold: fill_cols / ldo");
  if ((p->C & 7) == 0) ZS_REQUIRE(p->ldg % 8 == 0, "zs_conv2d_fold: ldg alignment");
  const int cols = p->out_f32 ? p->C : p->fill_cols;
  const int64_t per_row = (int64_t)p->Wd * ((cols + 7) / 8);
  ZS_REQUIRE(per_row < (1ll << 30) && (int64_t)p->B * p->H_in <= 65535, "zs_conv2d_fold: grid too large (B*H_in <= 65535)");
  const dim3 grid((unsigned)((per_row + NTP - 1) / NTP), (unsigned)(p->B * p->H_in));
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(conv2d_fold_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(conv2d_fold_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_conv2d_fold");
}

extern "C" int zs_conv2d_unpad(const ZsConv2dUnpad* p, void* stream) {
  ZS_REQUIRE(p && p->g00 && p->g01 && p->g10 && p->g11 && p->out, "zs_conv2d_unpad: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->pad >= 0 && p->Hp > p->pad && p->Wp > p->pad && p->Hp <= p->H + 2 * p->pad &&
                 p->Wp <= p->W + 2 * p->pad, "zs_conv2d_unpad: sizes");
  ZS_REQUIRE(p->pad_mode != ZS_PAD_REFLECT || (p->pad < p->H && p->pad < p->W), "zs_conv2d_unpad: Padding size should be less than the corresponding input dimension");
  ZS_REQUIRE(p->C % 8 == 0 && p->ldg % 8 == 0 && p->ldg >= p->C && p->ldo % 8 == 0 && p->fill_cols % 8 == 0 && p->fill_cols >= p->C && p->ldo >= p->fill_cols &&
                 al16p(p->g00) && al16p(p->g01) && al16p(p->g10) && al16p(p->g11) && al16p(p->out) && (!p->add || (al16p(p->add) && p->ldadd % 8 == 0)),
             "zs_conv2d_unpad: C, row pitches and fill_cols must be multiples of 8, operands 16-byte aligned");
  const int64_t total = (int64_t)p->B * p->H * p->W * (p->fill_cols / 8);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(conv2d_unpad_kernel<float>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(conv2d_unpad_kernel<bf16_t>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_conv2d_unpad");
}

extern "C" size_t zs_row_moments_workspace(int32_t B, int32_t T, int32_t C) {
  const size_t nslab = (size_t)((T + MOM_ROWS - 1) / MOM_ROWS);
  return (size_t)B * nslab * 3 * (size_t)C * sizeof(float);
}

extern "C" int zs_row_moments(const ZsRowMoments* p, void* stream) {
  ZS_REQUIRE(p && p->u && p->s1 && p->partial, "zs_row_moments: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->C > 0 && p->C % 8 == 0, "zs_row_moments: sizes (C %% 8 == 0)");
  ZS_REQUIRE(!p->s2 || p->v, "zs_row_moments: s2 needs v");
  ZS_REQUIRE(!p->s3 || p->w, "zs_row_moments: s3 needs w");
  ZS_REQUIRE(p->partial_bytes >= zs_row_moments_workspace(p->B, p->T, p->C), "zs_row_moments: workspace too small");
  ZS_REQUIRE(al16p(p->u) && p->ldu % 8 == 0 && (!p->v || (al16p(p->v) && p->ldv % 8 == 0)) && (!p->y || (al16p(p->y) && p->ldy % 8 == 0)) &&
                 (!p->w || (al16p(p->w) && p->ldw % 8 == 0)), "zs_row_moments: alignment");
  const int nslab = (p->T + MOM_ROWS - 1) / MOM_ROWS;
  ZS_REQUIRE(nslab <= 65535 && p->B <= 65535, "zs_row_moments: grid too large");
  dim3 grid((unsigned)((p->C + 63) / 64), (unsigned)p->B, (unsigned)nslab);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(row_moments_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, *p, nslab);
  else hipLaunchKernelGGL(row_moments_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, *p, nslab);
  int rc = zs_check_launch("zs_row_moments");
  if (rc) return rc;
  const int64_t n = (int64_t)p->B * p->C;
  hipLaunchKernelGGL(row_moments_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, nslab);
  return zs_check_launch("zs_row_moments.finish");
}

extern "C" int zs_in2d_stats(int32_t dtype, const void* y, int64_t ldy, int32_t B, int32_t T, int32_t C, float eps, float* mean, float* rstd,
                             float* partial, size_t partial_bytes, void* stream) {
  ZS_REQUIRE(y && mean && rstd && partial, "zs_in2d_stats: null operand");
  ZS_REQUIRE(dtype == ZS_F32 || dtype == ZS_BF16, "zs_in2d_stats: bad dtype");
  ZS_REQUIRE(B > 0 && T > 0 && C > 0 && C % 8 == 0 && al16p(y) && ldy % 8 == 0, "zs_in2d_stats: sizes / alignment (C %% 8 == 0)");
  ZS_REQUIRE(partial_bytes >= zs_row_moments_workspace(B, T, C), "zs_in2d_stats: workspace too small");
  const int nslab = (T + MOM_ROWS - 1) / MOM_ROWS;
  ZS_REQUIRE(nslab <= 65535 && B <= 65535, "zs_in2d_stats: grid too large");
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)B, (unsigned)nslab);
  if (dtype == ZS_F32) hipLaunchKernelGGL(in2d_stats_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, y, ldy, T, C, partial, nslab);
  else hipLaunchKernelGGL(in2d_stats_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, y, ldy, T, C, partial, nslab);
  int rc = zs_check_launch("zs_in2d_stats");
  if (rc) return rc;
  const int64_t n = (int64_t)B * C;
  hipLaunchKernelGGL(in2d_stats_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, B, C, nslab, eps, mean, rstd);
  return zs_check_launch("zs_in2d_stats.finish");
}

extern "C" int zs_in2d_finalize(const float* s1, const float* q, float* mean, float* rstd, int64_t n_bc, int32_t T, float eps, void* stream) {
  ZS_REQUIRE(s1 && q && mean && rstd && n_bc > 0 && T > 0, "zs_in2d_finalize: bad args");
  hipLaunchKernelGGL(in2d_finalize_kernel, dim3((unsigned)((n_bc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s1, q, mean, rstd, n_bc,
                     1.0f / (float)T, eps);
  return zs_check_launch("zs_in2d_finalize");
}

#define ZS_ROWS_OK(ptr, ld) (al16p(ptr) && (ld) % 8 == 0)

extern "C" int zs_in2d_fwd(const ZsIn2dFwd* p, void* stream) {
  ZS_REQUIRE(p && p->y && p->a && p->mean && p->rstd, "zs_in2d_fwd: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->C > 0 && ZS_ROWS_OK(p->y, p->ldy) && ZS_ROWS_OK(p->a, p->lda) && p->lda >= ((p->C + 7) / 8) * 8, "zs_in2d_fwd: sizes / alignment");
  ZS_REQUIRE(p->C <= 8 * NTP && p->B <= 65535, "zs_in2d_fwd: C <= %d, B <= 65535", 8 * NTP);
  const dim3 grid = row_walk_grid(p->B, p->T, p->C);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(in2d_fwd_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(in2d_fwd_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_in2d_fwd");
}

extern "C" int zs_in2d_bwd(const ZsIn2dBwd* p, void* stream) {
  ZS_REQUIRE(p && p->ga && p->a && p->y && p->S1 && p->S2 && p->rstd && p->gz, "zs_in2d_bwd: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->C > 0 && ZS_ROWS_OK(p->ga, p->ldga) && ZS_ROWS_OK(p->a, p->lda) && ZS_ROWS_OK(p->y, p->ldy) &&
                 ZS_ROWS_OK(p->gz, p->ldgz) && (!p->ga2 || ZS_ROWS_OK(p->ga2, p->ldga2)), "zs_in2d_bwd: sizes / alignment");
  ZS_REQUIRE(p->C <= 8 * NTP && p->B <= 65535, "zs_in2d_bwd: C <= %d, B <= 65535", 8 * NTP);
  const dim3 grid = row_walk_grid(p->B, p->T, p->C);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(in2d_bwd_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(in2d_bwd_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_in2d_bwd");
}

extern "C" int zs_in2d_adj(const ZsIn2dAdj* p, void* stream) {
  ZS_REQUIRE(p && p->gbz && p->y && p->a && p->ga && p->A1 && p->A2 && p->S2 && p->rstd && p->gba && p->xba, "zs_in2d_adj: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->C > 0 && ZS_ROWS_OK(p->gbz, p->ldgbz) && ZS_ROWS_OK(p->y, p->ldy) && ZS_ROWS_OK(p->a, p->lda) &&
                 ZS_ROWS_OK(p->ga, p->ldga) && ZS_ROWS_OK(p->gba, p->ldgba) && ZS_ROWS_OK(p->xba, p->ldxba), "zs_in2d_adj: sizes / alignment");
  ZS_REQUIRE(p->C <= 8 * NTP && p->B <= 65535, "zs_in2d_adj: C <= %d, B <= 65535", 8 * NTP);
  const dim3 grid = row_walk_grid(p->B, p->T, p->C);
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(in2d_adj_kernel<float>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(in2d_adj_kernel<bf16_t>, grid, dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_in2d_adj");
}

extern "C" int zs_lerp_rows(const float* x, const float* y, const float* alpha, float* out, int32_t B, int64_t n, void* stream) {
  ZS_REQUIRE(x && y && alpha && out && B > 0 && n > 0, "zs_lerp_rows: bad args");
  hipLaunchKernelGGL(lerp_rows_kernel, dim3(grid_for((int64_t)B * n)), dim3(NTP), 0, (hipStream_t)stream, x, y, alpha, out, (int)B, n);
  return zs_check_launch("zs_lerp_rows");
}

extern "C" int zs_gp_penalty(const float* g, int32_t B, int64_t n, float scale, float* s_out, float* gp_out, float* gbar, void* stream) {
  ZS_REQUIRE(g && s_out && B > 0 && n > 0, "zs_gp_penalty: bad args");
  hipLaunchKernelGGL(gp_norm_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, g, n, s_out);
  int rc = zs_check_launch("zs_gp_penalty.norm");
  if (rc) return rc;
  hipLaunchKernelGGL(gp_finish_kernel, dim3(grid_for(gbar ? (int64_t)B * n : 1)), dim3(NTP), 0, (hipStream_t)stream, g, (const float*)s_out, (int)B, n, scale,
                     gp_out, gbar);
  return zs_check_launch("zs_gp_penalty.finish");
}

extern "C" int zs_gen_combine_fwd(const float* xd, const float* m, int64_t ld_in, float* x_gen, int64_t rows, int32_t F, int32_t mode, void* stream) {
  ZS_REQUIRE(xd && m && x_gen && rows > 0 && F > 0 && ld_in >= F, "zs_gen_combine_fwd: bad args");
  hipLaunchKernelGGL(gen_combine_fwd_kernel, dim3(grid_for(rows * F)), dim3(NTP), 0, (hipStream_t)stream, xd, m, ld_in, x_gen, rows, (int)F, (int)mode);
  return zs_check_launch("zs_gen_combine_fwd");
}

extern "C" int zs_gen_combine_bwd(const ZsGenCombineBwd* p, void* stream) {
  ZS_REQUIRE(p && p->dx_gen && p->m && p->dpre && (!p->mode || p->xd), "zs_gen_combine_bwd: null operand");
  ZS_DT_OK(p);
  ZS_REQUIRE(p->rows > 0 && p->F > 0 && p->fill_cols >= p->F && p->fill_cols <= p->ldo && p->ld_in >= p->F && p->ld_dx >= p->F, "zs_gen_combine_bwd: sizes");
  const int64_t total = p->rows * p->fill_cols;
  if (p->dtype == ZS_F32) hipLaunchKernelGGL(gen_combine_bwd_kernel<float>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(gen_combine_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NTP), 0, (hipStream_t)stream, *p);
  return zs_check_launch("zs_gen_combine_bwd");
}

extern "C" int zs_l1_plain(const float* a, const float* b, int64_t n, float scale, float* partial, float* loss_out, float* d, void* stream) {
  ZS_REQUIRE(a && b && partial && loss_out && n > 0, "zs_l1_plain: bad args");
  int64_t nb = (n + 4095) / 4096;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(l1_plain_stage1, dim3((unsigned)nb), dim3(NTP), 0, (hipStream_t)stream, a, b, n, scale, partial, d);
  int rc = zs_check_launch("zs_l1_plain.stage1");
  if (rc) return rc;
  hipLaunchKernelGGL(l1_plain_stage2, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)partial, (int)nb, n, loss_out);
  return zs_check_launch("zs_l1_plain.stage2");
}
