// zs_norm.hip -- InstanceNorm1d (+dropout +residual) forward/backward and the backward-path
// "gradient combine" kernel.  All three share one decomposition of a channels-last [B,T,C] tensor:
// a 256-thread workgroup owns (sample b, 64 channels); thread (cg = tid&7, rg = tid>>3) owns
// channels 8cg..8cg+7 (one 16-byte access for bf16, two for fp32 -> 8 lanes cover one full 128-byte
// line of a row) of rows t = rg + 32*i.  The per-(b,c) reduction over T happens across the 32 row
// groups through LDS; values stay in registers between the passes (T <= 256 => <= 8 rows/thread), so
// each activation is read from HBM exactly once.  These kernels are HBM-bound.
#include <stdlib.h>

#include <atomic>

#include "zs_common.h"

namespace {

constexpr int RPT = 8;                                   // rows per thread
// The decomposition (threads NTN, channels per workgroup CHUNK = NTN/RG*8, row groups RG): 256 threads, 64 channels (128 B
// of a row per workgroup), 32 row groups -> T <= 256.  A 512-thread / 256-channel variant (512 B of a row per workgroup)
// measured slower in isolation (instnorm_bwd 62 vs 41 us at [256,128,1024]) and equal inside the step.
struct Narrow { static constexpr int NTN = 256, CHUNK = 64, RG = 32; };
// Short samples (T <= 64: the T' = 16 / 32 / 64 layers, 14 of the step's 17 norms): 8 row groups x 256 channels, so a thread still
// has up to 8 rows in flight and a workgroup moves 4x the bytes per barrier (Narrow at T = 32: one row per thread, 1.5 TB/s).
struct Wide8 { static constexpr int NTN = 256, CHUNK = 256, RG = 8; };
// The reduction over T has ONE order for every decomposition: 32 partials (rows r, r + 32, ... in ascending order -- Narrow's row
// groups), summed r = 0..31.  A thread of a decomposition with RG < 32 row groups carries NP = 32 / RG of those partials (its
// row i belongs to partial i % NP), so a sample's statistics do not depend on the launch's T (ragged batches) nor on the variant.
constexpr int RGN = 32;

// sum the per-thread partials pv[NP][8] of channel group cg over the 32 canonical row groups; result broadcast into out[8].
template <typename S>
__device__ __forceinline__ void colreduce(const float (&pv)[RGN / S::RG][8], float (&out)[8], float* red, float* tot, int cg, int rg, int tid) {
  constexpr int CHUNK = S::CHUNK, RG = S::RG, NP = RGN / RG;
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(rg + RG * j) * CHUNK + cg * 8 + k] = pv[j][k];
  __syncthreads();
  for (int c = tid; c < CHUNK; c += S::NTN) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RGN; ++r) s += red[r * CHUNK + c];
    tot[c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) out[k] = tot[cg * 8 + k];
  __syncthreads();
}

__device__ __forceinline__ float keep_scale(const uint8_t* mask, int64_t mask_ld, float p, uint64_t seed, uint32_t sid,
                                            int64_t row, int c, int C, float inv_keep) {
  if (p <= 0.f) return 1.f;
  bool keep;
  if (mask) keep = (c < mask_ld) ? (mask[row * mask_ld + c] != 0) : true;
  else keep = zs_keep(seed, sid, (uint64_t)row * (uint64_t)C + (uint64_t)c, p);
  return keep ? inv_keep : 0.f;
}

template <typename T, typename S>
__global__ __launch_bounds__(S::NTN) void instnorm_fwd_kernel(const ZsInstNormFwd p) {
  constexpr int CHUNK = S::CHUNK, RG = S::RG, CGN = S::CHUNK / 8, NP = RGN / RG;
  __shared__ float red[RGN * CHUNK];
  __shared__ float tot[CHUNK];
  const int tid = threadIdx.x, cg = tid % CGN, rg = tid / CGN;
  const int b = blockIdx.y, c0 = blockIdx.x * CHUNK + cg * 8;
  const bool cvalid = c0 < p.C;
  // ragged batch: this sample's own length (p.T / p.T_res stay the row strides); same row -> thread mapping and reduction order
  // as a launch with T = Tb, so a sample's result does not depend on what it is batched with
  const int Tb = p.lengths ? p.lengths[b] : p.T;
  const int Tres_b = p.res_lengths ? p.res_lengths[b] : p.T_res;
  const T* x = (const T*)p.x;
  Raw8<T> v[RPT];            // rows stay as loaded (bf16: 4 registers per 8 values instead of 8) between the passes: occupancy
  float s[NP][8];
  // the residual rows are fetched together with x (raw: they are not needed before the second pass), so that their HBM
  // latency is not paid again behind the two reductions
  const T* res = (const T*)p.res;
  Raw8<T> rr0[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    rr0[i].zero();
    if (cvalid && t < Tb) {
      if (p.res_mode == ZS_RES_IDENTITY) rr0[i].ld(res + ((int64_t)b * p.T + t) * p.ldres + c0);
      else if (p.res_mode == ZS_RES_UPSAMPLE2) rr0[i].ld(res + ((int64_t)b * p.T_res + (t >> 1)) * p.ldres + c0);
      else if (p.res_mode == ZS_RES_AVGPOOL2) {
        // F.pad(x, (0, T%2), reflect|constant) + avg_pool1d(2)   (model/model.py:424-425)
        rr0[i].ld(res + ((int64_t)b * p.T_res + 2 * t) * p.ldres + c0);     // the odd partner row is fetched in the second pass
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) s[j][k] = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    v[i].zero();
    if (cvalid && t < Tb) v[i].ld(x + ((int64_t)b * p.T + t) * p.ldx + c0);
  }
  float mean[8], rstd[8];
  const float invT = 1.f / (float)Tb;
  if (p.stats_given) {                                                         // (kernel-uniform) statistics from the caller
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mean[k] = cvalid ? p.mean[(int64_t)b * p.C + c0 + k] : 0.f;
      rstd[k] = cvalid ? p.rstd[(int64_t)b * p.C + c0 + k] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float f[8]; v[i].cvt(f);                    // rows past T are zeros
#pragma unroll
      for (int k = 0; k < 8; ++k) s[i % NP][k] += f[k];
    }
    float tsum[8];
    colreduce<S>(s, tsum, red, tot, cg, rg, tid);
#pragma unroll
    for (int k = 0; k < 8; ++k) mean[k] = tsum[k] * invT;
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[j][k] = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int t = rg + RG * i;
      if (t < Tb) {
        float f[8]; v[i].cvt(f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = f[k] - mean[k]; s[i % NP][k] += d * d; }
      }
    }
    colreduce<S>(s, tsum, red, tot, cg, rg, tid);
#pragma unroll
    for (int k = 0; k < 8; ++k) rstd[k] = 1.f / sqrtf(tsum[k] * invT + p.eps);
  }
  if (!cvalid) return;
  if (p.mean && rg == 0 && !p.stats_given) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { p.mean[(int64_t)b * p.C + c0 + k] = mean[k]; p.rstd[(int64_t)b * p.C + c0 + k] = rstd[k]; }
  }
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
  float e2[8];
  if (p.out2) {
    const int64_t vi = p.idx ? p.idx[b] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) e2[k] = (p.vec2 && (c0 + k) < p.vec2_cols) ? p.vec2[vi * p.vec2_ld + c0 + k] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    if (t >= p.T) continue;
    const int64_t row = (int64_t)b * p.T + t;
    float o[8], f[8];
    if (t >= Tb) {                                 // ragged batch: rows past the sample's length are zeros
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = 0.f;
      store8<T>((T*)p.out + row * p.ldo + c0, o);
      if (p.out2) store8<T>((T*)p.out2 + row * p.ldo2 + c0, o);
      continue;
    }
    v[i].cvt(f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (f[k] - mean[k]) * rstd[k];
      o[k] *= keep_scale(p.mask, p.mask_ld, p.drop_p, seed, p.stream_id, row, c0 + k, p.C, inv_keep);
    }
    if (p.res_mode == ZS_RES_IDENTITY || p.res_mode == ZS_RES_UPSAMPLE2) {
      float r[8]; rr0[i].cvt(r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r[k];
    } else if (p.res_mode == ZS_RES_AVGPOOL2) {
      float r0[8], r1[8]; rr0[i].cvt(r0);
      int t1 = 2 * t + 1; bool have = true;
      if (t1 >= Tres_b) { if (p.res_pad_mode == ZS_PAD_REFLECT) t1 = Tres_b - 2; else have = false; }
      if (have) load8<T>(res + ((int64_t)b * p.T_res + t1) * p.ldres + c0, r1);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) r1[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += (r0[k] + r1[k]) * 0.5f;
    }
    store8<T>((T*)p.out + row * p.ldo + c0, o);
    if (p.out2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += e2[k];
      store8<T>((T*)p.out2 + row * p.ldo2 + c0, o);
    }
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(S::NTN) void instnorm_bwd_kernel(const ZsInstNormBwd p) {
  constexpr int CHUNK = S::CHUNK, RG = S::RG, CGN = S::CHUNK / 8, NP = RGN / RG;
  __shared__ float red[RGN * CHUNK];
  __shared__ float tot[CHUNK];
  const int tid = threadIdx.x, cg = tid % CGN, rg = tid / CGN;
  const int b = blockIdx.y, c0 = blockIdx.x * CHUNK + cg * 8;
  const bool cvalid = c0 < p.C;
  float mean[8], rstd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mean[k] = cvalid ? p.mean[(int64_t)b * p.C + c0 + k] : 0.f;
    rstd[k] = cvalid ? p.rstd[(int64_t)b * p.C + c0 + k] : 0.f;
  }
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
  Raw8<T> g[RPT], xv[RPT];      // as loaded: 8 instead of 16 registers per row pair in bf16 (this kernel was at 256 VGPRs)
  float pg[NP][8], pgx[NP][8], sg[8], sgx[8];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) { pg[j][k] = 0.f; pgx[j][k] = 0.f; }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    g[i].zero(); xv[i].zero();
    if (cvalid && t < p.T) {
      const int64_t row = (int64_t)b * p.T + t;
      g[i].ld((const T*)p.dout + row * p.ldd + c0);
      xv[i].ld((const T*)p.x + row * p.ldx + c0);
    }
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    if (cvalid && t < p.T) {
      const int64_t row = (int64_t)b * p.T + t;
      float gf[8], xf[8];
      g[i].cvt(gf); xv[i].cvt(xf);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        gf[k] *= keep_scale(p.mask, p.mask_ld, p.drop_p, seed, p.stream_id, row, c0 + k, p.C, inv_keep);
        const float xh = (xf[k] - mean[k]) * rstd[k];
        pg[i % NP][k] += gf[k]; pgx[i % NP][k] += gf[k] * xh;
      }
    }
  }
  colreduce<S>(pg, sg, red, tot, cg, rg, tid);
  colreduce<S>(pgx, sgx, red, tot, cg, rg, tid);
  if (!cvalid) return;
  const float invT = 1.f / (float)p.T;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    if (t >= p.T) continue;
    const int64_t row = (int64_t)b * p.T + t;
    float o[8], gf[8], xf[8];
    g[i].cvt(gf); xv[i].cvt(xf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gf[k] *= keep_scale(p.mask, p.mask_ld, p.drop_p, seed, p.stream_id, row, c0 + k, p.C, inv_keep);   // same mask as in the first pass
      const float xh = (xf[k] - mean[k]) * rstd[k];
      const float dx = rstd[k] * (gf[k] - sg[k] * invT - xh * (sgx[k] * invT));
      o[k] = dx * dlrelu_f(xf[k], p.slope);
    }
    store8<T>((T*)p.dz + row * p.ldz + c0, o);
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(S::NTN) void grad_combine_kernel(const ZsGradCombine p) {
  constexpr int CHUNK = S::CHUNK, RG = S::RG, CGN = S::CHUNK / 8, NP = RGN / RG;
  __shared__ float red[RGN * CHUNK];
  __shared__ float tot[CHUNK];
  const int tid = threadIdx.x, cg = tid % CGN, rg = tid / CGN;
  const int b = blockIdx.y, c0 = blockIdx.x * CHUNK + cg * 8;
  const bool cvalid = c0 < p.C;
  const int Tp = p.T + p.pad_left + p.pad_right;
  const T* gp = (const T*)p.gp;
  float v[RPT][8];
  float ps[NP][8], s[8];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) ps[j][k] = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
    if (cvalid && t < p.T) {
      const T* base = gp + (int64_t)b * Tp * p.ldg + c0;
      load8<T>(base + (int64_t)(t + p.pad_left) * p.ldg, v[i]);
      if (p.pad_mode == ZS_PAD_REFLECT) {
        if (t >= 1 && t <= p.pad_left) {                       // left reflection partner
          float r[8]; load8<T>(base + (int64_t)(p.pad_left - t) * p.ldg, r);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[i][k] += r[k];
        }
        if (t <= p.T - 2 && t >= p.T - 1 - p.pad_right) {      // right reflection partner
          float r[8]; load8<T>(base + (int64_t)(p.pad_left + 2 * (p.T - 1) - t) * p.ldg, r);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[i][k] += r[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) ps[i % NP][k] += v[i][k];
    }
  }
  if (p.emb_sum) {                                             // uniform branch
    colreduce<S>(ps, s, red, tot, cg, rg, tid);
    if (cvalid && rg == 0) {                                   // this workgroup owns (b, these channels): plain RMW
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c0 + k < p.emb_cols) p.emb_sum[(int64_t)b * p.emb_ld + c0 + k] += s[k];
    }
  }
  if (!cvalid || p.out == nullptr) return;
  const T* res = (const T*)p.res;
  const T* da = (const T*)p.dact_src;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int t = rg + RG * i;
    if (t >= p.T) continue;
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = v[i][k];
    if (p.res_mode == ZS_RES_IDENTITY) {
      float r[8]; load8<T>(res + ((int64_t)b * p.T + t) * p.ldres + c0, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r[k];
    } else if (p.res_mode == ZS_RES_AVGPOOL2) {                // backward of avg_pool1d(2), even T
      float r[8]; load8<T>(res + ((int64_t)b * (p.T >> 1) + (t >> 1)) * p.ldres + c0, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += 0.5f * r[k];
    } else if (p.res_mode == ZS_RES_UPSAMPLE2) {               // backward of nearest x2
      float r0[8], r1[8];
      load8<T>(res + ((int64_t)b * 2 * p.T + 2 * t) * p.ldres + c0, r0);
      load8<T>(res + ((int64_t)b * 2 * p.T + 2 * t + 1) * p.ldres + c0, r1);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r0[k] + r1[k];
    }
    int64_t orow = (int64_t)b * p.T + t; int ocol = c0;
    if (p.unshuffle) { orow = (int64_t)b * (p.T >> 1) + (t >> 1); ocol = (t & 1) * p.C + c0; }
    if (da) {
      float y[8]; load8<T>(da + orow * p.dact_ld + ocol, y);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] *= dlrelu_f(y[k], p.slope);
    }
    store8<T>((T*)p.out + orow * p.ldo + ocol, o);
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

std::atomic<int> g_norm_wide{1};
// longest T that takes the 8x256 decomposition in instnorm_fwd, instnorm_bwd, grad_combine ("norm_lim0..2" knobs: tools/instnorm_probe.py)
int g_norm_lim[3] = {64, 32, 64};
// Wide8 for short samples of wide layers in bf16 (fp32 rows are twice the registers; it is the parity dtype, not the fast one)
// Measured at B = 256, C = 1024, bf16 (tools/instnorm_probe.py; us for 32x64 | 8x256):
//   forward   T = 16: 29.2 | 11.9   32: 31.4 | 17.0   64: 36.6 | 26.2
//   backward  T = 16: 18.4 | 14.7   32: 19.7 | 19.1   64: 25.4 | 30.7
//   combine   T = 16:  7.7 |  7.0   32: 11.6 |  9.1   64: 18.0 | 16.0
// A 16x128 decomposition for T <= 128 won 10 % on the forward kernel alone (50.1 -> 44.9 us) and nothing in the step
// (11.05-11.07 against 11.01-11.03 ms with 8x256 only, same box): not kept.
// 8x256 for short samples of wide layers in bf16 (fp32 rows are twice the registers; it is the parity dtype, not the fast one)
bool norm_wide(int dtype, int T, int C, int max_T) {
  return g_norm_wide.load(std::memory_order_relaxed) != 0 && dtype == ZS_BF16 && T <= max_T && T <= Wide8::RG * RPT && C >= Wide8::CHUNK;
}

#define ZS_NORM_LAUNCH(KERNEL, MAXT)                                                                                        \
  do {                                                                                                                      \
    if (norm_wide(p->dtype, p->T, p->C, MAXT)) {                                                                            \
      dim3 grid((p->C + Wide8::CHUNK - 1) / Wide8::CHUNK, p->B);                                                            \
      hipLaunchKernelGGL((KERNEL<bf16_t, Wide8>), grid, dim3(Wide8::NTN), 0, (hipStream_t)stream, *p);                      \
    } else {                                                                                                                \
      dim3 grid((p->C + Narrow::CHUNK - 1) / Narrow::CHUNK, p->B);                                                          \
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((KERNEL<float, Narrow>), grid, dim3(Narrow::NTN), 0, (hipStream_t)stream, *p); \
      else hipLaunchKernelGGL((KERNEL<bf16_t, Narrow>), grid, dim3(Narrow::NTN), 0, (hipStream_t)stream, *p);                \
    }                                                                                                                       \
  } while (0)

}  // namespace

int zs_norm_lim_option(int i, int value) { const int o = g_norm_lim[i]; g_norm_lim[i] = value; return o; }
// "norm_wide" knob of zs_set_option
int zs_norm_wide_option(int value) { return g_norm_wide.exchange(value ? 1 : 0, std::memory_order_relaxed); }

extern "C" int zs_instnorm_fwd(const ZsInstNormFwd* p, void* stream) {
  ZS_REQUIRE(p && p->x && p->out, "zs_instnorm_fwd: null operand");
  ZS_REQUIRE(!p->stats_given || (p->mean && p->rstd), "zs_instnorm_fwd: stats_given needs mean and rstd");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_instnorm_fwd: bad dtype");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->T <= Narrow::RG * RPT && p->C > 0 && p->C % 8 == 0, "zs_instnorm_fwd: need 0<T<=256 (T=%d), C%%8==0 (C=%d)", p->T, p->C);
  ZS_REQUIRE(al16(p->x) && al16(p->out) && (p->ldx * es) % 16 == 0 && (p->ldo * es) % 16 == 0, "zs_instnorm_fwd: alignment");
  ZS_REQUIRE(!p->out2 || (al16(p->out2) && (p->ldo2 * es) % 16 == 0 && (!p->vec2 || p->idx)), "zs_instnorm_fwd: out2");
  ZS_REQUIRE((p->mean == nullptr) == (p->rstd == nullptr), "zs_instnorm_fwd: mean/rstd");
  if (p->res_mode != ZS_RES_NONE) {
    ZS_REQUIRE(p->res && al16(p->res) && (p->ldres * es) % 16 == 0, "zs_instnorm_fwd: residual");
    if (p->res_mode == ZS_RES_UPSAMPLE2) ZS_REQUIRE(p->T_res * 2 == p->T, "zs_instnorm_fwd: upsample T_res %d vs T %d", p->T_res, p->T);
    if (p->res_mode == ZS_RES_AVGPOOL2) ZS_REQUIRE((p->T_res + 1) / 2 == p->T && p->T_res >= 2, "zs_instnorm_fwd: avgpool T_res %d vs T %d", p->T_res, p->T);
  }
  ZS_REQUIRE(p->drop_p >= 0.f && p->drop_p < 1.f, "zs_instnorm_fwd: drop_p");
  ZS_NORM_LAUNCH(instnorm_fwd_kernel, g_norm_lim[0]);
  return zs_check_launch("zs_instnorm_fwd");
}

extern "C" int zs_instnorm_bwd(const ZsInstNormBwd* p, void* stream) {
  ZS_REQUIRE(p && p->dout && p->x && p->mean && p->rstd && p->dz, "zs_instnorm_bwd: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_instnorm_bwd: bad dtype");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->T <= Narrow::RG * RPT && p->C > 0 && p->C % 8 == 0, "zs_instnorm_bwd: sizes");
  ZS_REQUIRE(al16(p->dout) && al16(p->x) && al16(p->dz) && (p->ldd * es) % 16 == 0 && (p->ldx * es) % 16 == 0 && (p->ldz * es) % 16 == 0,
             "zs_instnorm_bwd: alignment");
  ZS_NORM_LAUNCH(instnorm_bwd_kernel, g_norm_lim[1]);
  return zs_check_launch("zs_instnorm_bwd");
}

extern "C" int zs_grad_combine(const ZsGradCombine* p, void* stream) {
  ZS_REQUIRE(p && p->gp && (p->out || p->emb_sum), "zs_grad_combine: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_grad_combine: bad dtype");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->T <= Narrow::RG * RPT && p->C > 0 && p->C % 8 == 0, "zs_grad_combine: sizes T=%d C=%d", p->T, p->C);
  ZS_REQUIRE(p->pad_left >= 0 && p->pad_right >= 0 && (p->pad_mode != ZS_PAD_REFLECT || (p->pad_left < p->T && p->pad_right < p->T)),
             "zs_grad_combine: pads");
  ZS_REQUIRE(al16(p->gp) && (p->ldg * es) % 16 == 0 && (!p->out || (al16(p->out) && (p->ldo * es) % 16 == 0)), "zs_grad_combine: alignment");
  if (p->res_mode != ZS_RES_NONE) {
    ZS_REQUIRE(p->res && al16(p->res) && (p->ldres * es) % 16 == 0, "zs_grad_combine: residual");
    ZS_REQUIRE(p->res_mode != ZS_RES_AVGPOOL2 || p->T % 2 == 0, "zs_grad_combine: avgpool backward needs even T");
  }
  ZS_REQUIRE(!p->unshuffle || (p->T % 2 == 0 && (p->C * es) % 16 == 0), "zs_grad_combine: unshuffle");
  ZS_REQUIRE(!p->dact_src || (al16(p->dact_src) && (p->dact_ld * es) % 16 == 0), "zs_grad_combine: dact");
  ZS_NORM_LAUNCH(grad_combine_kernel, g_norm_lim[2]);
  return zs_check_launch("zs_grad_combine");
}
