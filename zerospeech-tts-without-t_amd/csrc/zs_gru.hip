// zs_gru.hip -- bidirectional GRU (model/model.py:59-66; nn.GRU gate order r,z,n, zero h0).
// The input projections for all T and both directions are one zs_gemm_conv call made by the caller.
// Here: the sequential part.  Per time step, ONE grouped MFMA product (both directions as groups,
// h_{t-1} W_hh^T, fp32 out) and ONE gate kernel are enqueued; the loop over T lives in the library so
// the host pays one C call per layer, and the whole layer is graph-capturable.
//   forward :  r = s(gi_r + gh_r + b_hr)  z = s(gi_z + gh_z + b_hz)  n = tanh(gi_n + r*(gh_n + b_hn))
//              h = (1-z)*n + z*h_prev
//   backward:  BPTT with dh carried as (direct part dh*z) + (dgh W_hh) computed by the same product kernel.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "zs_common.h"

extern "C" size_t zs_gru_work_bytes(int32_t B, int32_t H);

namespace {

// The work buffer starts with a 256-byte header whose first word is the error word of the persistent kernels (set when a bounded
// spin timed out; read by zs_gru_check): header and exchange granules are zeroed by ONE memset per launch.
constexpr size_t GRU_WORK_HDR = 256;

constexpr int NTG = 256;

struct GateFwdArgs {
  const void* gi; int64_t ldgi;
  const float* gh;        // [2][B][3H]
  const float* bhh; int64_t bhh_gstride;
  float* hstate;          // [2][B][H]
  void* out; int64_t ldo; int out_col;
  void* gates;            // [B][T][2][4H] or null
  int B, T, H, step;
  int dir1_forward;       // ZsGruFwd.dir1_forward: direction 1 also runs t = 0 .. T-1
};

template <typename T>
__global__ void gru_gate_fwd_kernel(const GateFwdArgs a) {
  const int64_t total = (int64_t)2 * a.B * a.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / ((int64_t)a.B * a.H));
    const int64_t rem = i - (int64_t)d * a.B * a.H;
    const int b = (int)(rem / a.H), j = (int)(rem - (int64_t)b * a.H);
    const int t = (d == 0 || a.dir1_forward) ? a.step : a.T - 1 - a.step;
    const int H = a.H;
    const T* gi = (const T*)a.gi + ((int64_t)b * a.T + t) * a.ldgi + (int64_t)d * 3 * H;
    const float* bh = a.bhh + (int64_t)d * a.bhh_gstride;
    float ghr = 0.f, ghz = 0.f, ghn = 0.f, hp = 0.f;
    if (a.step > 0) {
      const float* gh = a.gh + ((int64_t)d * a.B + b) * 3 * H;
      ghr = gh[j]; ghz = gh[H + j]; ghn = gh[2 * H + j];
      hp = a.hstate[((int64_t)d * a.B + b) * H + j];
    }
    const float r = 1.f / (1.f + expf(-(Elem<T>::ld(gi + j) + (ghr + bh[j]))));
    const float z = 1.f / (1.f + expf(-(Elem<T>::ld(gi + H + j) + (ghz + bh[H + j]))));
    const float hn = ghn + bh[2 * H + j];
    const float n = tanhf(Elem<T>::ld(gi + 2 * H + j) + r * hn);
    const float h = (1.f - z) * n + z * hp;
    a.hstate[((int64_t)d * a.B + b) * H + j] = h;
    Elem<T>::st((T*)a.out + ((int64_t)b * a.T + t) * a.ldo + a.out_col + d * H + j, h);
    if (a.gates) {
      T* gs = (T*)a.gates + (((int64_t)b * a.T + t) * 2 + d) * 4 * H;
      Elem<T>::st(gs + j, r); Elem<T>::st(gs + H + j, z); Elem<T>::st(gs + 2 * H + j, n); Elem<T>::st(gs + 3 * H + j, hn);
    }
  }
}

struct GateBwdArgs {
  const void* dout; int64_t ldd; int dout_col;
  const void* out; int64_t ldo; int out_col;
  const void* gates;
  float* dhd;             // [2][B][H] direct carry  dh*z
  const float* dhg;       // [2][B][H] product carry dgh W_hh
  void* dgi; int64_t ldgi;
  void* dgh; int64_t ldgh;
  int B, T, H, step;
};

template <typename T>
__global__ void gru_gate_bwd_kernel(const GateBwdArgs a) {
  const int64_t total = (int64_t)2 * a.B * a.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / ((int64_t)a.B * a.H));
    const int64_t rem = i - (int64_t)d * a.B * a.H;
    const int b = (int)(rem / a.H), j = (int)(rem - (int64_t)b * a.H);
    const int H = a.H;
    const int t = d == 0 ? a.T - 1 - a.step : a.step;          // reverse of the forward order
    const int64_t row = (int64_t)b * a.T + t;
    float dh = Elem<T>::ld((const T*)a.dout + row * a.ldd + a.dout_col + d * H + j);
    const int64_t ci = ((int64_t)d * a.B + b) * H + j;
    if (a.step > 0) dh += a.dhd[ci] + a.dhg[ci];
    const T* gs = (const T*)a.gates + (row * 2 + d) * 4 * H;
    const float r = Elem<T>::ld(gs + j), z = Elem<T>::ld(gs + H + j), n = Elem<T>::ld(gs + 2 * H + j), hn = Elem<T>::ld(gs + 3 * H + j);
    float hp = 0.f;
    if (a.step < a.T - 1) {
      const int tp = d == 0 ? t - 1 : t + 1;
      hp = Elem<T>::ld((const T*)a.out + ((int64_t)b * a.T + tp) * a.ldo + a.out_col + d * H + j);
    }
    const float dn = dh * (1.f - z);
    const float dz = dh * (hp - n);
    a.dhd[ci] = dh * z;
    const float dn_pre = dn * (1.f - n * n);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    const float dz_pre = dz * z * (1.f - z);
    T* gi = (T*)a.dgi + row * a.ldgi + (int64_t)d * 3 * H;
    T* gh = (T*)a.dgh + row * a.ldgh + (int64_t)d * 3 * H;
    Elem<T>::st(gi + j, dr_pre); Elem<T>::st(gi + H + j, dz_pre); Elem<T>::st(gi + 2 * H + j, dn_pre);
    Elem<T>::st(gh + j, dr_pre); Elem<T>::st(gh + H + j, dz_pre); Elem<T>::st(gh + 2 * H + j, dn_pre * r);
  }
}


// ------------------------------------------------------------------------------------------------
// Small-M product for the recurrence: out[M, N] = A[M, K] * W[N, K]^T with M = batch (a few hundred rows).
// One workgroup owns 16 rows x (16*NT16) columns and its 4 waves split K (one LDS reduce at the end); operands
// are loaded as MFMA fragments straight from global memory (L2/MALL resident: h_{t-1} and W_hh are re-read
// every step), 16 B per lane per fragment, so the serial chain of a step is K/4 deep and 2 workgroups/CU run.  bf16: v_mfma_f32_16x16x32_bf16;
// fp32: four v_mfma_f32_16x16x4_f32 per 16-byte fragment (lane q supplies k = 4q+s in step s on both sides).
// GRU = true fuses the forward gate math: W rows are packed [r(32) | z(32) | n(32)] per 32 hidden units, so
// a lane's accumulators (col = lane&15) of tiles {0,1}/{2,3}/{4,5} are r/z/n of the SAME (row, unit).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct RowGemmArgs {
  const void* A; int64_t a_row_stride; int64_t a_gstride;
  const void* W; int64_t ldw; int64_t w_gstride;
  int M, N, K;
  float* out; int64_t out_gstride;     // plain mode: fp32 [g][M][N]
  GateFwdArgs gate;                    // MODE 1: fused forward gates
  GateBwdArgs gbw;                     // MODE 2: fused backward gates of the NEXT BPTT step
};

template <typename T> struct Frag16;
template <> struct Frag16<bf16_t> {
  static constexpr int KSTEP = 32;     // elements of K per 16-byte-per-lane fragment step
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Frag16<float> {
  static constexpr int KSTEP = 16;
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

constexpr int RB = 2;                                  // 16-row blocks per workgroup (32 rows): W is re-read B/32 times per step

template <typename T, int NT16, int MODE>   // MODE 0 plain fp32 out, 1 GRU forward gates, 2 GRU backward gates (next step)
__global__ __launch_bounds__(256) void rowblock_gemm_kernel(const RowGemmArgs a) {
  constexpr bool GRU = (MODE == 1);
  // workgroup = one (16*RB)-row x (16*NT16)-column tile; its 4 waves split K and reduce through LDS
  constexpr int KSTEP = Frag16<T>::KSTEP;
  constexpr int EPL = 16 / (int)sizeof(T);            // elements per lane per fragment
  constexpr int ROWS = 16 * RB;
  __shared__ float part[4][RB][NT16][4][64];           // [wave][row block][tile][reg][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int g = blockIdx.z;
  const int m0 = blockIdx.y * ROWS;
  const int n0 = blockIdx.x * (16 * NT16);
  const bool skip = GRU && (a.gate.step == 0);         // h_{-1} = 0

  // ---- GRU epilogue operands: issue their loads first so they fly under the K loop ----
  // thread -> (row, unit) pairs: unit u = tid & 31, rows e_row + 8*pz
  constexpr int NP = ROWS / 8;
  const int eu = tid & 31, e_row = tid >> 5;
  float e_gi[NP][3], e_hp[NP], e_b[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int pz = 0; pz < NP; ++pz) { e_gi[pz][0] = e_gi[pz][1] = e_gi[pz][2] = 0.f; e_hp[pz] = 0.f; }
  int e_t = 0;
  if constexpr (GRU) {
    const GateFwdArgs& ga = a.gate;
    const int d = g, H = ga.H, j = blockIdx.x * 32 + eu;
    e_t = (d == 0 || ga.dir1_forward) ? ga.step : ga.T - 1 - ga.step;
    const float* bh = ga.bhh + (int64_t)d * ga.bhh_gstride;
    e_b[0] = bh[j]; e_b[1] = bh[H + j]; e_b[2] = bh[2 * H + j];
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {
      const int b = m0 + e_row + 8 * pz;
      if (b < a.M) {
        const T* gi = (const T*)ga.gi + ((int64_t)b * ga.T + e_t) * ga.ldgi + (int64_t)d * 3 * H;
        e_gi[pz][0] = Elem<T>::ld(gi + j); e_gi[pz][1] = Elem<T>::ld(gi + H + j); e_gi[pz][2] = Elem<T>::ld(gi + 2 * H + j);
        if (!skip) e_hp[pz] = ga.hstate[((int64_t)d * ga.B + b) * H + j];
      }
    }
  }

  // ---- MODE 2: operands of the next BPTT step's gate math (same (row, unit) ownership), loaded up front ----
  float w_r[NP], w_z[NP], w_n[NP], w_hn[NP], w_dh[NP], w_hp[NP];
  int w_t = 0;
  if constexpr (MODE == 2) {
    static_assert(NT16 == 2 || MODE != 2, "fused backward gates need a 32-column tile");
    const GateBwdArgs& gb = a.gbw;
    const int d = g, H = gb.H, j = blockIdx.x * 32 + eu;
    w_t = d == 0 ? gb.T - 1 - gb.step : gb.step;              // time index of BPTT step gb.step
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {
      w_r[pz] = w_z[pz] = w_n[pz] = w_hn[pz] = w_dh[pz] = w_hp[pz] = 0.f;
      const int b = m0 + e_row + 8 * pz;
      if (b < a.M) {
        const int64_t row = (int64_t)b * gb.T + w_t;
        const T* gs = (const T*)gb.gates + (row * 2 + d) * 4 * H;
        w_r[pz] = Elem<T>::ld(gs + j); w_z[pz] = Elem<T>::ld(gs + H + j); w_n[pz] = Elem<T>::ld(gs + 2 * H + j); w_hn[pz] = Elem<T>::ld(gs + 3 * H + j);
        w_dh[pz] = Elem<T>::ld((const T*)gb.dout + row * gb.ldd + gb.dout_col + d * H + j) + gb.dhd[((int64_t)d * gb.B + b) * H + j];
        if (gb.step < gb.T - 1) {
          const int tp = d == 0 ? w_t - 1 : w_t + 1;
          w_hp[pz] = Elem<T>::ld((const T*)gb.out + ((int64_t)b * gb.T + tp) * gb.ldo + gb.out_col + d * H + j);
        }
      }
    }
  }

  f32x4_t acc[RB][NT16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int c = 0; c < NT16; ++c) acc[rb][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (!skip) {
    const T* Arow[RB];
    bool rvalid[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int row = m0 + 16 * rb + r;
      rvalid[rb] = row < a.M;
      Arow[rb] = (const T*)a.A + (int64_t)g * a.a_gstride + (int64_t)(rvalid[rb] ? row : 0) * a.a_row_stride + q * EPL;
    }
    const T* Wrow = (const T*)a.W + (int64_t)g * a.w_gstride + (int64_t)(n0 + r) * a.ldw + q * EPL;
    const int nks = a.K / KSTEP;
    const int per = (nks + 3) >> 2;
    int ks = wave * per;
    const int kend = min(nks, ks + per);
    for (; ks + 4 <= kend; ks += 4) {                  // 4 k-steps of fragments in flight before the first MFMA
      uint4 fa[4][RB], fb[4][NT16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fa[u][rb] = *reinterpret_cast<const uint4*>(Arow[rb] + (ks + u) * KSTEP);
#pragma unroll
        for (int c = 0; c < NT16; ++c) fb[u][c] = *reinterpret_cast<const uint4*>(Wrow + (int64_t)(16 * c) * a.ldw + (ks + u) * KSTEP);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (!rvalid[rb]) fa[u][rb] = make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int c = 0; c < NT16; ++c) Frag16<T>::mma(fa[u][rb], fb[u][c], acc[rb][c]);
        }
    }
    for (; ks < kend; ++ks) {
      uint4 fa[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        fa[rb] = *reinterpret_cast<const uint4*>(Arow[rb] + ks * KSTEP);
        if (!rvalid[rb]) fa[rb] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < NT16; ++c) {
        const uint4 fb = *reinterpret_cast<const uint4*>(Wrow + (int64_t)(16 * c) * a.ldw + ks * KSTEP);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) Frag16<T>::mma(fa[rb], fb, acc[rb][c]);
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int c = 0; c < NT16; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][rb][c][i][lane] = acc[rb][c][i];
  __syncthreads();
  // element (row rr in [0,ROWS), col cc) of tile c lives at reg i = rr & 3 of lane ((rr & 15) >> 2) * 16 + cc, block rr >> 4
  auto total = [&](int c, int rr, int cc) -> float {
    const int rb = rr >> 4, l = ((rr & 15) >> 2) * 16 + cc, i = rr & 3;
    return (part[0][rb][c][i][l] + part[1][rb][c][i][l]) + (part[2][rb][c][i][l] + part[3][rb][c][i][l]);
  };
  if constexpr (MODE == 2) {
    // dh_{s+1} = dout + dh_s*z_s (dhd) + dgh_s W_hh (this product); gate backward of step s+1 for the owned (row, unit)
    const GateBwdArgs& gb = a.gbw;
    const int d = g, H = gb.H, j = blockIdx.x * 32 + eu;
    const int hh = eu >> 4, cc = eu & 15;
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {
      const int rr = e_row + 8 * pz;
      const int b = m0 + rr;
      if (b >= a.M) continue;
      const float dh = w_dh[pz] + total(hh, rr, cc);
      const float r_ = w_r[pz], z_ = w_z[pz], n_ = w_n[pz];
      const float dn = dh * (1.f - z_);
      const float dz = dh * (w_hp[pz] - n_);
      gb.dhd[((int64_t)d * gb.B + b) * H + j] = dh * z_;
      const float dn_pre = dn * (1.f - n_ * n_);
      const float dr_pre = dn_pre * w_hn[pz] * r_ * (1.f - r_);
      const float dz_pre = dz * z_ * (1.f - z_);
      const int64_t row = (int64_t)b * gb.T + w_t;
      T* gi = (T*)gb.dgi + row * gb.ldgi + (int64_t)d * 3 * H;
      T* gh = (T*)gb.dgh + row * gb.ldgh + (int64_t)d * 3 * H;
      Elem<T>::st(gi + j, dr_pre); Elem<T>::st(gi + H + j, dz_pre); Elem<T>::st(gi + 2 * H + j, dn_pre);
      Elem<T>::st(gh + j, dr_pre); Elem<T>::st(gh + H + j, dz_pre); Elem<T>::st(gh + 2 * H + j, dn_pre * r_);
    }
  } else if constexpr (!GRU) {
    float* out = a.out + (int64_t)g * a.out_gstride;
    constexpr int NC = 16 * NT16;
    for (int e = tid; e < ROWS * NC; e += 256) {        // coalesced along the columns
      const int rr = e / NC, col = e - rr * NC;
      const int m = m0 + rr;
      if (m < a.M) out[(int64_t)m * a.N + n0 + col] = total(col >> 4, rr, col & 15);
    }
  } else {
    const GateFwdArgs& ga = a.gate;
    const int d = g, H = ga.H, j = blockIdx.x * 32 + eu;
    const int hh = eu >> 4, cc = eu & 15;
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {
      const int rr = e_row + 8 * pz;
      const int b = m0 + rr;
      if (b >= a.M) continue;
      const float ghr = total(hh, rr, cc), ghz = total(2 + hh, rr, cc), ghn = total(4 + hh, rr, cc);
      const float rg = 1.f / (1.f + expf(-(e_gi[pz][0] + (ghr + e_b[0]))));
      const float zg = 1.f / (1.f + expf(-(e_gi[pz][1] + (ghz + e_b[1]))));
      const float hn = ghn + e_b[2];
      const float ng = tanhf(e_gi[pz][2] + rg * hn);
      const float h = (1.f - zg) * ng + zg * e_hp[pz];
      ga.hstate[((int64_t)d * ga.B + b) * H + j] = h;
      Elem<T>::st((T*)ga.out + ((int64_t)b * ga.T + e_t) * ga.ldo + ga.out_col + d * H + j, h);
      if (ga.gates) {
        T* gs = (T*)ga.gates + (((int64_t)b * ga.T + e_t) * 2 + d) * 4 * H;
        Elem<T>::st(gs + j, rg); Elem<T>::st(gs + H + j, zg); Elem<T>::st(gs + 2 * H + j, ng); Elem<T>::st(gs + 3 * H + j, hn);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent recurrence: ONE launch per layer and direction pair; the loop over time runs inside the kernel.
// Workgroup (hc, rb, dir) owns 32 hidden units x 32 batch rows for all T steps.  Its slice of W_hh (96 rows x H,
// 96 KiB in bf16 at H = 512) is loaded into registers once (each of the 4 waves keeps its quarter of K); per step it
// only reads h_{t-1} of its 32 rows (written one step earlier by the H/32 workgroups of the same (rb, dir) group), runs
// the MFMAs, reduces over the 4 waves through LDS, does the gate math for its (row, unit) elements (h_prev stays in a
// register) and publishes its 32x32 slice of h_t.
// Hand-off between the workgroups of a group: DATA-TAGGED GRANULES, no flag, no fence, no drain (cdna guide, recipe R2).
// h travels through an exchange buffer hx[dir][step parity][row][granule] of naturally aligned 8-byte words
// {payload: 4 bytes of h (2 bf16 / 1 fp32), tag: step + 1}, each written by ONE agent-scope (sc1, write-through) store.
// A consumer wave sweeps the granules of its own MFMA fragments with sc1 loads (they bypass this CU's L1) until every
// tag equals the step it needs; the payloads ARE the fragment.  Two parities suffice: nobody can write h_{s+1} before
// every workgroup of the group has published h_s, i.e. has finished reading h_{s-1}.  hx is zeroed before every launch
// (tags of an earlier call must not match).  Spins are bounded: on timeout the error word is set and the wave stops
// waiting (results are then wrong, never a hang).  The whole grid (<= one workgroup per CU) is co-resident: host check.
// An earlier version with write-through payload + arrival counter + poll measured 13.7 us per step against 11.1 us for
// one launch per step (the drain -> counter -> poll -> re-read chain is longer than a kernel boundary).
// ------------------------------------------------------------------------------------------------
struct GruPersistArgs {
  const void* gi; int64_t ldgi;
  const void* whh; int64_t ldw; int64_t w_gstride;
  const float* bhh; int64_t bhh_gstride;
  void* out; int64_t ldo; int out_col;
  void* gates;
  unsigned long long* hx;        // exchange granules [2 dirs][2 parities][rows_pad][gpr]; zeroed before the launch
  unsigned* err;                 // error word (behind hx), zeroed with it
  unsigned* status;              // optional caller-owned STICKY status word: bit 0 is OR-ed in on a timeout, never cleared here
  unsigned spin_limit;
  int B, T, H, rows_pad;
  const float* bcast_vec; int64_t bcast_ld; const int64_t* bcast_idx; int bcast_col;   // ZsGruFwd.bcast_*
  int dir1_forward;              // ZsGruFwd.dir1_forward
};

// 16-byte agent-scope (sc1: bypasses this CU's L1) load.  Inline asm so that all loads of a sweep are in flight together
// (hipcc issues relaxed atomic loads one at a time, each behind a wait); the caller waits with s_waitcnt vmcnt(0).
typedef __attribute__((ext_vector_type(4))) unsigned gu32x4_t;
__device__ __forceinline__ void load16_sc1_issue(gu32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned tag, unsigned payload) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> struct Pair;
template <> struct Pair<bf16_t> {
  typedef unsigned raw_t;            // two bf16 as loaded: converting at the load site would park the wave on the HBM latency
  static __device__ __forceinline__ raw_t ld_raw(const bf16_t* p) { return *reinterpret_cast<const unsigned*>(p); }
  static __device__ __forceinline__ void cvt(raw_t w, float& a, float& b) { a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ void ld(const bf16_t* p, float& a, float& b) { cvt(ld_raw(p), a, b); }
  static __device__ __forceinline__ unsigned pack(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
  static __device__ __forceinline__ void st(bf16_t* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = pack(a, b); }
  // the pair (unit j, j+1) of a row is ONE granule
  static __device__ __forceinline__ void publish(unsigned long long* row, int j, unsigned tag, float a, float b) {
    store_granule(row + (j >> 1), tag, pack(a, b));
  }
};
template <> struct Pair<float> {
  typedef float2 raw_t;
  static __device__ __forceinline__ raw_t ld_raw(const float* p) { return *reinterpret_cast<const float2*>(p); }
  static __device__ __forceinline__ void cvt(raw_t v, float& a, float& b) { a = v.x; b = v.y; }
  static __device__ __forceinline__ void ld(const float* p, float& a, float& b) { cvt(ld_raw(p), a, b); }
  static __device__ __forceinline__ void st(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
  static __device__ __forceinline__ void publish(unsigned long long* row, int j, unsigned tag, float a, float b) {
    store_granule(row + j, tag, __float_as_uint(a));
    store_granule(row + j + 1, tag, __float_as_uint(b));
  }
};

// The same load in a form the compiler can see (raw buffer load, cache policy sc1, volatile): it inserts the counted waits
// itself, so such loads may stay in flight across any amount of code -- the wide kernels request the next step's granules right
// after their own hand-off.  (The inline-asm form above is only safe when the wait follows the issue directly: a register the
// compiler believes defined may be copied before the data has arrived.)
typedef __amdgpu_buffer_rsrc_t gru_rsrc_t;
__device__ __forceinline__ gru_rsrc_t make_buffer_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ gu32x4_t load16_sc1(gru_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(gu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, (int)(0x80000000u | 16u)));
}

constexpr unsigned GRU_SPIN_LIMIT = 1u << 21;   // default of the "gru_spin_limit" option

template <typename T, int PER>     // PER = k-steps (of Frag16<T>::KSTEP) per wave: H = 4 * PER * KSTEP
__global__ __launch_bounds__(256) void gru_persist_fwd_kernel(const GruPersistArgs a) {
  constexpr int KSTEP = Frag16<T>::KSTEP;
  constexpr int EPL = 16 / (int)sizeof(T);
  constexpr int GPE = 4 / (int)sizeof(T);              // elements per granule payload
  __shared__ float part[4][RB][6][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int d = blockIdx.z, H = a.H, T_ = a.T;
  const int m0 = blockIdx.y * 16 * RB;
  const int gpr = H / GPE;                             // granules per row
  const int64_t par_stride = (int64_t)a.rows_pad * gpr;
  unsigned long long* hx_d = a.hx + (int64_t)d * 2 * par_stride;

  // W_hh slice of this workgroup, this wave's K quarter: registers for the whole sequence
  uint4 fb[PER][6];
  {
    const T* Wrow = (const T*)a.whh + (int64_t)d * a.w_gstride + (int64_t)(blockIdx.x * 96 + r) * a.ldw + q * EPL;
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
      for (int c = 0; c < 6; ++c) fb[u][c] = *reinterpret_cast<const uint4*>(Wrow + (int64_t)(16 * c) * a.ldw + (wave * PER + u) * KSTEP);
  }
  // fragment granules of this lane: row m0 + 16 rb + r, elements (wave*PER + u)*KSTEP + q*EPL .. +EPL-1 = 4 granules
  const unsigned long long* Ag[RB];
  bool rvalid[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = m0 + 16 * rb + r;
    rvalid[rb] = row < a.B;
    Ag[rb] = hx_d + (int64_t)row * gpr + (q * EPL + wave * PER * KSTEP) / GPE;
  }
  // epilogue ownership: unit pair (2*eu2, 2*eu2+1) of the 32 units, rows e_row + 16*pz
  const int eu2 = tid & 15, e_row = tid >> 4;
  const int j = blockIdx.x * 32 + 2 * eu2;
  const int hh = (2 * eu2) >> 4, cc = (2 * eu2) & 15;
  const float* bh = a.bhh + (int64_t)d * a.bhh_gstride;
  const float b_r0 = bh[j], b_r1 = bh[j + 1], b_z0 = bh[H + j], b_z1 = bh[H + j + 1], b_n0 = bh[2 * H + j], b_n1 = bh[2 * H + j + 1];
  float hp[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  bool dead = false;                                   // a sweep timed out: stop waiting (wave-uniform)
  float bc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};           // broadcast values of this thread's (row, unit pair): constant over time
  if (a.bcast_vec) {
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int b = m0 + e_row + 16 * pz;
      if (b < a.B) { const float* v = a.bcast_vec + a.bcast_idx[b] * a.bcast_ld + d * H + j; bc[pz][0] = v[0]; bc[pz][1] = v[1]; }
    }
  }

  const bool fwd_t = d == 0 || a.dir1_forward;         // this direction walks t = 0 .. T-1
  for (int s = 0; s < T_; ++s) {
    const int t = fwd_t ? s : T_ - 1 - s;
    // gate inputs of this step do not depend on other workgroups: issue them before the sweep
    typename Pair<T>::raw_t q_r[2], q_z[2], q_n[2];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int b = min(m0 + e_row + 16 * pz, a.B - 1);      // rows past B: any valid address, the values are not used
      const T* gi = (const T*)a.gi + ((int64_t)b * T_ + t) * a.ldgi + (int64_t)d * 3 * H + j;
      q_r[pz] = Pair<T>::ld_raw(gi); q_z[pz] = Pair<T>::ld_raw(gi + H); q_n[pz] = Pair<T>::ld_raw(gi + 2 * H);
    }
    f32x4_t acc[RB][6];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[rb][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // sweep the granules of h_{s-1} (tag s, parity (s-1)&1) until every tag of this wave's fragments matches
      const int64_t poff = (int64_t)((s - 1) & 1) * par_stride;
      gu32x4_t g0[PER][RB], g1[PER][RB];
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int u = 0; u < PER; ++u)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const unsigned long long* gp = Ag[rb] + poff + (u * KSTEP) / GPE;
            load16_sc1_issue(g0[u][rb], gp);
            load16_sc1_issue(g1[u][rb], gp + 2);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);             // nothing that reads the granules may be scheduled above the wait
        bool ok = true;
#pragma unroll
        for (int u = 0; u < PER; ++u)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            asm volatile("" : "+v"(g0[u][rb]), "+v"(g1[u][rb]));       // the values are defined only from here on
            const bool m = (g0[u][rb].y == (unsigned)s) & (g0[u][rb].w == (unsigned)s) & (g1[u][rb].y == (unsigned)s) & (g1[u][rb].w == (unsigned)s);
            ok &= (m | !rvalid[rb]);
          }
        if (__all(ok) || dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) {
          if (lane == 0) {
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.status) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          dead = true;
          break;
        }
      }
#pragma unroll
      for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          uint4 fa = make_uint4(g0[u][rb].x, g0[u][rb].z, g1[u][rb].x, g1[u][rb].z);
          if (!rvalid[rb]) fa = make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 6; ++c) Frag16<T>::mma(fa, fb[u][c], acc[rb][c]);
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][rb][c][i][lane] = acc[rb][c][i];
    __syncthreads();
    auto total = [&](int c, int rr, int col) -> float {
      const int rb = rr >> 4, l = ((rr & 15) >> 2) * 16 + col, i = rr & 3;
      return (part[0][rb][c][i][l] + part[1][rb][c][i][l]) + (part[2][rb][c][i][l] + part[3][rb][c][i][l]);
    };
    unsigned long long* hx_w = hx_d + (int64_t)(s & 1) * par_stride;
    // Gate math of both row groups first, then both hand-off granules, then the ordinary stores: vmcnt counts loads and stores
    // in one in-order counter, so a wait for this step's gate inputs placed behind a store (as the compiler must assume when
    // loads are consumed after stores were issued) is a wait for that store's round trip -- on the critical path of the group.
    float hv[2][2], rg[2][2], zg[2][2], ng[2][2], hn[2][2];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int rr = e_row + 16 * pz;
      float gi_r[2], gi_z[2], gi_n[2];
      Pair<T>::cvt(q_r[pz], gi_r[0], gi_r[1]); Pair<T>::cvt(q_z[pz], gi_z[0], gi_z[1]); Pair<T>::cvt(q_n[pz], gi_n[0], gi_n[1]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float ghr = total(hh, rr, cc + e), ghz = total(2 + hh, rr, cc + e), ghn = total(4 + hh, rr, cc + e);
        rg[pz][e] = 1.f / (1.f + expf(-(gi_r[e] + (ghr + (e ? b_r1 : b_r0)))));
        zg[pz][e] = 1.f / (1.f + expf(-(gi_z[e] + (ghz + (e ? b_z1 : b_z0)))));
        hn[pz][e] = ghn + (e ? b_n1 : b_n0);
        ng[pz][e] = tanhf(gi_n[e] + rg[pz][e] * hn[pz][e]);
        hv[pz][e] = (1.f - zg[pz][e]) * ng[pz][e] + zg[pz][e] * hp[pz][e];
        hp[pz][e] = hv[pz][e];
      }
    }
    if (s + 1 < T_) {                                  // first: the others wait for these
#pragma unroll
      for (int pz = 0; pz < 2; ++pz) {
        const int b = m0 + e_row + 16 * pz;
        if (b < a.B) Pair<T>::publish(hx_w + (int64_t)b * gpr, j, (unsigned)(s + 1), hv[pz][0], hv[pz][1]);
      }
    }
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int b = m0 + e_row + 16 * pz;
      if (b < a.B) {
        Pair<T>::st((T*)a.out + ((int64_t)b * T_ + t) * a.ldo + a.out_col + d * H + j, hv[pz][0], hv[pz][1]);
        if (a.gates) {
          T* gs = (T*)a.gates + (((int64_t)b * T_ + t) * 2 + d) * 4 * H + j;
          Pair<T>::st(gs, rg[pz][0], rg[pz][1]); Pair<T>::st(gs + H, zg[pz][0], zg[pz][1]);
          Pair<T>::st(gs + 2 * H, ng[pz][0], ng[pz][1]); Pair<T>::st(gs + 3 * H, hn[pz][0], hn[pz][1]);
        }
        if (a.bcast_vec) Pair<T>::st((T*)a.out + ((int64_t)b * T_ + t) * a.ldo + a.bcast_col + d * H + j, bc[pz][0], bc[pz][1]);
      }
    }
    __syncthreads();                                   // `part` is free for the next step
  }
}

// ------------------------------------------------------------------------------------------------
// "Wide" persistent recurrence (bf16, H = 32 * KS in {128, 256, 512}): the same data-tagged granule hand-off, re-tiled so that
// a time step has ONE workgroup barrier, no cross-wave reduction and a third of the exchange partners' data per wave.
// A group of H/64 workgroups owns 16 batch rows of one direction for the whole sequence; workgroup `hc` owns 64 hidden units,
// each of its 4 waves (one per SIMD: a 512-VGPR budget) 16 units x 3 gates over the FULL K = H -- its slice of W_hh, 3*KS
// 16-byte fragments (192 VGPRs at H = 512), stays in registers.  Per step:
//   sweep: the 256 threads fetch the granules of the other workgroups' h_{s-1} (16 rows x (H-64) units, KS/2-1 16-byte sc1
//          loads per thread) until every tag matches and drop the payloads into an LDS image of h_{s-1} (the own 64 units were
//          put there at the end of the previous step) | barrier | KS x 3 MFMAs 16x16x32 per wave, A fragments from LDS |
//   gate math in the accumulator layout (row 4*(lane>>4)+i, unit lane&15: r, z, n of one unit are in the same lane) |
//   publish h_s (granules, pairs of units through a DPP swap), then the ordinary stores, then the own units into the OTHER LDS image.
// Two LDS images by step parity make the single barrier sufficient (the image written at the end of step s was last read before
// the barrier of step s).  Against the 32x32 tiling above (6.3 us per step at H = 512): see tools/gru_bench.py.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void gru_wide_fwd_kernel(const GruPersistArgs a) {
  constexpr int H = 32 * KS;
  constexpr int NCH = KS / 2 - 1;                       // sweep loads per thread: 16 rows x (H - 64) units / 4 units / 256
  constexpr int NCHA = NCH > 0 ? NCH : 1;
  constexpr int CPR = (H - 64) / 4;                     // 16-byte chunks (4 units) of the other workgroups per row
  constexpr int PITCH = H * 2 + 16;                     // bytes per row of the LDS image
  __shared__ __attribute__((aligned(16))) unsigned char himg[2][16 * PITCH];      // h_{s-1}, by step parity
  __shared__ __attribute__((aligned(16))) bf16_t gi_s[2][16][3][64];               // gate inputs of the step, by step parity
  __shared__ __attribute__((aligned(16))) bf16_t out_s[2][5][16][64];              // h, r, z, n, hn of the step, by step parity
  __shared__ __attribute__((aligned(16))) bf16_t bc_s[16][64];                     // broadcast values (constant over time)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int d = blockIdx.z, T_ = a.T;
  const int m0 = blockIdx.y * 16;
  const int u_wg = blockIdx.x * 64;                     // first unit of this workgroup
  const int ul = wave * 16 + r;                         // this lane's unit inside the workgroup (accumulator layout)
  const int unit = u_wg + ul;
  constexpr int gpr = H / 2;                            // granules per row
  const int64_t par_stride = (int64_t)a.rows_pad * gpr;
  unsigned long long* hx_d = a.hx + (int64_t)d * 2 * par_stride;
  const gru_rsrc_t hx_rs = make_buffer_rsrc(hx_d, (unsigned)(2 * par_stride * 8));     // both parities of this direction

  // W_hh rows of this wave: packed (gate-interleaved per 32 units) row of (gate g, unit u) = (u/32)*96 + g*32 + u%32
  uint4 fb[KS][3];
  {
    const int u0 = u_wg + wave * 16;
    const bf16_t* Wb = (const bf16_t*)a.whh + (int64_t)d * a.w_gstride + (int64_t)((u0 >> 5) * 96 + (u0 & 31) + r) * a.ldw + q * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int g = 0; g < 3; ++g) fb[ks][g] = *reinterpret_cast<const uint4*>(Wb + (int64_t)(32 * g) * a.ldw + ks * 32);
  }
  const float* bh = a.bhh + (int64_t)d * a.bhh_gstride;
  const float b_r = bh[unit], b_z = bh[H + unit], b_n = bh[2 * H + unit];
  // sweep chunks of this thread: chunk c = tid + 256 i -> (row, 4 units of another workgroup)
  int ch_row[NCHA], ch_unit[NCHA];
  unsigned ch_off[NCHA];                                 // byte offset of the chunk's granules in one parity of hx_d
  bool ch_valid[NCHA];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + 256 * i;
    const int row = c / CPR, cu = (c - row * CPR) * 4;
    ch_row[i] = row; ch_unit[i] = cu < u_wg ? cu : cu + 64;
    ch_valid[i] = (m0 + row) < a.B;
    ch_off[i] = (unsigned)(((int64_t)(m0 + row) * gpr + (ch_unit[i] >> 1)) * 8);
  }
  const unsigned par_bytes = (unsigned)(par_stride * 8);
  // cooperative 16-byte transfers (8 units each).  gate inputs: 16 rows x 3 gates x 8 chunks = 384 (thread: tid, tid + 256 < 384);
  // outputs: 6 arrays (h, r, z, n, hn, broadcast) x 16 rows x 8 chunks = 768 (thread: tid + 256 k, k < 3)
  auto gi_src = [&](int id, int t) -> const bf16_t* {
    const int row = id / 24, rem = id - row * 24, g = rem >> 3, c = rem & 7;
    const int b = min(m0 + row, a.B - 1);
    return (const bf16_t*)a.gi + ((int64_t)b * T_ + t) * a.ldgi + (int64_t)d * 3 * H + g * H + u_wg + 8 * c;
  };
  auto gi_dst = [&](int par, int id) -> uint4* {
    const int row = id / 24, rem = id - row * 24, g = rem >> 3, c = rem & 7;
    return reinterpret_cast<uint4*>(&gi_s[par][row][g][8 * c]);
  };
  auto copy_out = [&](int par, int t) {                 // staged outputs of one step -> global, 16-byte stores
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int id = tid + 256 * k;
      const int arr = id >> 7, rem = id & 127, row = rem >> 3, c = rem & 7;
      const int b = m0 + row;
      if (b >= a.B) continue;
      if (arr == 0) {
        *reinterpret_cast<uint4*>((bf16_t*)a.out + ((int64_t)b * T_ + t) * a.ldo + a.out_col + d * H + u_wg + 8 * c) =
            *reinterpret_cast<const uint4*>(&out_s[par][0][row][8 * c]);
      } else if (arr < 5) {
        if (a.gates)
          *reinterpret_cast<uint4*>((bf16_t*)a.gates + (((int64_t)b * T_ + t) * 2 + d) * 4 * H + (arr - 1) * H + u_wg + 8 * c) =
              *reinterpret_cast<const uint4*>(&out_s[par][arr][row][8 * c]);
      } else if (a.bcast_vec) {
        *reinterpret_cast<uint4*>((bf16_t*)a.out + ((int64_t)b * T_ + t) * a.ldo + a.bcast_col + d * H + u_wg + 8 * c) =
            *reinterpret_cast<const uint4*>(&bc_s[row][8 * c]);
      }
    }
  };
  float hp[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bcast_vec) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = min(m0 + 4 * q + i, a.B - 1);
      bc_s[4 * q + i][ul] = f2bf(a.bcast_vec[a.bcast_idx[b] * a.bcast_ld + d * H + unit]);
    }
  }
  const bool fwd_t = d == 0 || a.dir1_forward;         // this direction walks t = 0 .. T-1
  // gate inputs of step 0 (later steps: fetched one step ahead)
  uint4 gq0 = *reinterpret_cast<const uint4*>(gi_src(tid, fwd_t ? 0 : T_ - 1)), gq1 = make_uint4(0, 0, 0, 0);
  if (tid < 128) gq1 = *reinterpret_cast<const uint4*>(gi_src(tid + 256, fwd_t ? 0 : T_ - 1));
  gu32x4_t g4[NCHA];                                      // sweep loads of the coming step, issued right after the hand-off
  bool dead = false;

  for (int s = 0; s < T_; ++s) {
    unsigned char* const img = &himg[s & 1][0];
    f32x4_t acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if constexpr (NCH > 0) {
        // the granules of h_{s-1} (tag s, parity (s-1)&1) were requested at the end of the previous step
        const unsigned poff = (unsigned)((s - 1) & 1) * par_bytes;
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < NCH; ++i) ok &= (((g4[i].y == (unsigned)s) & (g4[i].w == (unsigned)s)) | !ch_valid[i]);
          if (__all(ok) || dead) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > a.spin_limit) {
            if (lane == 0) {
              __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (a.status) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            dead = true;
            break;
          }
#pragma unroll
          for (int i = 0; i < NCH; ++i) g4[i] = load16_sc1(hx_rs, poff + ch_off[i]);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          *reinterpret_cast<uint2*>(img + ch_row[i] * PITCH + ch_unit[i] * 2) = make_uint2(ch_valid[i] ? g4[i].x : 0u, ch_valid[i] ? g4[i].z : 0u);
      }
    }
    *gi_dst(s & 1, tid) = gq0;
    if (tid < 128) *gi_dst(s & 1, tid + 256) = gq1;
    __syncthreads();                                     // images of h_{s-1} and of this step's gate inputs are complete
    if (s > 0) copy_out((s - 1) & 1, fwd_t ? s - 1 : T_ - s);       // the previous step's outputs leave as 16-byte stores
    if (s + 1 < T_) {                                    // next step's gate inputs: in flight under this step
      const int tn = fwd_t ? s + 1 : T_ - 2 - s;
      gq0 = *reinterpret_cast<const uint4*>(gi_src(tid, tn));
      if (tid < 128) gq1 = *reinterpret_cast<const uint4*>(gi_src(tid + 256, tn));
    }
    if (s > 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 fa = *reinterpret_cast<const uint4*>(img + r * PITCH + q * 16 + ks * 64);
#pragma unroll
        for (int g = 0; g < 3; ++g) Frag16<bf16_t>::mma(fa, fb[ks][g], acc[g]);
      }
    }
    // gate math: lane owns (row 4q + i, unit) for i < 4
    float hv[4], rg[4], zg[4], ng[4], hn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi_r = bf2f(gi_s[s & 1][4 * q + i][0][ul]), gi_z = bf2f(gi_s[s & 1][4 * q + i][1][ul]), gi_n = bf2f(gi_s[s & 1][4 * q + i][2][ul]);
      rg[i] = 1.f / (1.f + expf(-(gi_r + (acc[0][i] + b_r))));
      zg[i] = 1.f / (1.f + expf(-(gi_z + (acc[1][i] + b_z))));
      hn[i] = acc[2][i] + b_n;
      ng[i] = tanhf(gi_n + rg[i] * hn[i]);
      hv[i] = (1.f - zg[i]) * ng[i] + zg[i] * hp[i];
      hp[i] = hv[i];
    }
    if (s + 1 < T_) {                                    // hand-off first: the group waits for it.  Even lanes publish (unit, unit+1)
      unsigned long long* hx_w = hx_d + (int64_t)(s & 1) * par_stride;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nb = __shfl_xor(hv[i], 1, 64);
        const int b = m0 + 4 * q + i;
        if (!(lane & 1) && b < a.B) store_granule(hx_w + (int64_t)b * gpr + (unit >> 1), (unsigned)(s + 1), Pair<bf16_t>::pack(hv[i], nb));
      }
      if constexpr (NCH > 0) {                           // ... and ask for the others' h_s at once: the round trip runs under the rest
        const unsigned poff = (unsigned)(s & 1) * par_bytes;
#pragma unroll
        for (int i = 0; i < NCH; ++i) g4[i] = load16_sc1(hx_rs, poff + ch_off[i]);
      }
      unsigned char* const nimg = &himg[(s + 1) & 1][0];
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16_t*>(nimg + (4 * q + i) * PITCH + unit * 2) = f2bf(hv[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out_s[s & 1][0][4 * q + i][ul] = f2bf(hv[i]);
      out_s[s & 1][1][4 * q + i][ul] = f2bf(rg[i]); out_s[s & 1][2][4 * q + i][ul] = f2bf(zg[i]);
      out_s[s & 1][3][4 * q + i][ul] = f2bf(ng[i]); out_s[s & 1][4][4 * q + i][ul] = f2bf(hn[i]);
    }
  }
  __syncthreads();
  copy_out((T_ - 1) & 1, fwd_t ? T_ - 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// Persistent BPTT, same hand-off: workgroup (hc, rb, dir) owns 32 hidden units x 32 rows.  Per step it needs dgh of the
// previous BPTT step for its rows and ALL 3H gate columns (published by the H/32 workgroups of its group as granules),
// multiplies by its 32 rows of W_hh^T (32 x 3H, register-resident: each wave keeps its quarter of K), adds dout and its
// own direct carry dh*z (a register) and runs the gate backward for its (row, unit) elements.
// ------------------------------------------------------------------------------------------------
struct GruPersistBwdArgs {
  const void* dout; int64_t ldd; int dout_col;
  const void* out; int64_t ldo; int out_col;
  const void* gates;
  const void* whh_t; int64_t ldw; int64_t w_gstride;
  void* dgi; int64_t ldgi;
  void* dgh; int64_t ldgh;
  unsigned long long* dx;        // exchange granules [2 dirs][2 parities][rows_pad][3H / GPE]; zeroed before the launch
  unsigned* err;
  unsigned* status;              // sticky status word (bit 1 = BPTT timeout), see GruPersistArgs
  unsigned spin_limit;
  int B, T, H, rows_pad;
};

// Tiling: 64 hidden units x 16 rows per workgroup (forward: 32 x 32): the BPTT sweep reads all 3H columns of its rows, so fewer
// rows per workgroup halve the granule traffic (31 MB per time step with 32 rows); W_hh^T slice = 64 x 3H = 192 VGPRs per lane.
constexpr int BNT = 4;             // 16-column MFMA tiles per workgroup (64 units)
template <typename T, int PERB>    // PERB = k-steps per wave: 3H = 4 * PERB * KSTEP
__global__ __launch_bounds__(256) void gru_persist_bwd_kernel(const GruPersistBwdArgs a) {
  constexpr int KSTEP = Frag16<T>::KSTEP;
  constexpr int EPL = 16 / (int)sizeof(T);
  constexpr int GPE = 4 / (int)sizeof(T);
  __shared__ float part[4][BNT][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int d = blockIdx.z, H = a.H, T_ = a.T;
  const int m0 = blockIdx.y * 16;
  const int gpr = 3 * H / GPE;
  const int64_t par_stride = (int64_t)a.rows_pad * gpr;
  unsigned long long* dx_d = a.dx + (int64_t)d * 2 * par_stride;

  uint4 fb[PERB][BNT];
  {
    const T* Wrow = (const T*)a.whh_t + (int64_t)d * a.w_gstride + (int64_t)(blockIdx.x * 16 * BNT + r) * a.ldw + q * EPL;
#pragma unroll
    for (int u = 0; u < PERB; ++u)
#pragma unroll
      for (int c = 0; c < BNT; ++c) fb[u][c] = *reinterpret_cast<const uint4*>(Wrow + (int64_t)(16 * c) * a.ldw + (wave * PERB + u) * KSTEP);
  }
  const bool rvalid = (m0 + r) < a.B;
  const unsigned long long* Ag = dx_d + (int64_t)(m0 + r) * gpr + (q * EPL + wave * PERB * KSTEP) / GPE;
  // gate-math ownership: unit pair (2*eu2, 2*eu2+1) of the 64 units, rows e_row + 8*pz
  const int eu2 = tid & 31, e_row = tid >> 5;
  const int j = blockIdx.x * 16 * BNT + 2 * eu2;
  const int hh = (2 * eu2) >> 4, cc = (2 * eu2) & 15;
  float dhd[2][2] = {{0.f, 0.f}, {0.f, 0.f}};          // direct carry dh * z of the previous BPTT step
  bool dead = false;

  for (int s = 0; s < T_; ++s) {
    const int t = d == 0 ? T_ - 1 - s : s;              // reverse of the forward order
    // operands of this step's gate math that no other workgroup writes in this launch: issue before the sweep
    typename Pair<T>::raw_t q_r[2], q_z[2], q_n[2], q_hn[2], q_do[2], q_hp[2];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int b = min(m0 + e_row + 8 * pz, a.B - 1);       // rows past B: any valid address, the values are not used
      const int64_t row = (int64_t)b * T_ + t;
      const T* gs = (const T*)a.gates + (row * 2 + d) * 4 * H + j;
      q_r[pz] = Pair<T>::ld_raw(gs); q_z[pz] = Pair<T>::ld_raw(gs + H); q_n[pz] = Pair<T>::ld_raw(gs + 2 * H); q_hn[pz] = Pair<T>::ld_raw(gs + 3 * H);
      q_do[pz] = Pair<T>::ld_raw((const T*)a.dout + row * a.ldd + a.dout_col + d * H + j);
      const int tp = s < T_ - 1 ? (d == 0 ? t - 1 : t + 1) : t;       // last BPTT step: h_prev = 0 (handled below)
      q_hp[pz] = Pair<T>::ld_raw((const T*)a.out + ((int64_t)b * T_ + tp) * a.ldo + a.out_col + d * H + j);
    }
    f32x4_t acc[BNT];
#pragma unroll
    for (int c = 0; c < BNT; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const int64_t poff = (int64_t)((s - 1) & 1) * par_stride;
      gu32x4_t g0[PERB], g1[PERB];
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int u = 0; u < PERB; ++u) {
          const unsigned long long* gp = Ag + poff + (u * KSTEP) / GPE;
          load16_sc1_issue(g0[u], gp);
          load16_sc1_issue(g1[u], gp + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < PERB; ++u) {
          asm volatile("" : "+v"(g0[u]), "+v"(g1[u]));
          const bool m = (g0[u].y == (unsigned)s) & (g0[u].w == (unsigned)s) & (g1[u].y == (unsigned)s) & (g1[u].w == (unsigned)s);
          ok &= (m | !rvalid);
        }
        if (__all(ok) || dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) {
          if (lane == 0) {
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.status) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          dead = true;
          break;
        }
      }
#pragma unroll
      for (int u = 0; u < PERB; ++u) {
        uint4 fa = make_uint4(g0[u].x, g0[u].z, g1[u].x, g1[u].z);
        if (!rvalid) fa = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < BNT; ++c) Frag16<T>::mma(fa, fb[u][c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < BNT; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][c][i][lane] = acc[c][i];
    __syncthreads();
    auto total = [&](int c, int rr, int col) -> float {      // row rr (0..15), column col of tile c
      const int l = (rr >> 2) * 16 + col, i = rr & 3;
      return (part[0][c][i][l] + part[1][c][i][l]) + (part[2][c][i][l] + part[3][c][i][l]);
    };
    unsigned long long* dx_w = dx_d + (int64_t)(s & 1) * par_stride;
    // gate backward of both row groups, then all hand-off granules, then the ordinary stores (see the forward kernel)
    float dr_pre[2][2], dz_pre[2][2], dn_pre[2][2], dnr[2][2];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int rr = e_row + 8 * pz;
      float w_r[2], w_z[2], w_n[2], w_hn[2], w_do[2], w_hp[2];
      Pair<T>::cvt(q_r[pz], w_r[0], w_r[1]); Pair<T>::cvt(q_z[pz], w_z[0], w_z[1]); Pair<T>::cvt(q_n[pz], w_n[0], w_n[1]);
      Pair<T>::cvt(q_hn[pz], w_hn[0], w_hn[1]); Pair<T>::cvt(q_do[pz], w_do[0], w_do[1]); Pair<T>::cvt(q_hp[pz], w_hp[0], w_hp[1]);
      if (s == T_ - 1) w_hp[0] = w_hp[1] = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float dh = w_do[e] + dhd[pz][e];
        if (s > 0) dh += total(hh, rr, cc + e);
        const float r_ = w_r[e], z_ = w_z[e], n_ = w_n[e];
        const float dn = dh * (1.f - z_);
        const float dz = dh * (w_hp[e] - n_);
        dhd[pz][e] = dh * z_;
        dn_pre[pz][e] = dn * (1.f - n_ * n_);
        dr_pre[pz][e] = dn_pre[pz][e] * w_hn[e] * r_ * (1.f - r_);
        dz_pre[pz][e] = dz * z_ * (1.f - z_);
        dnr[pz][e] = dn_pre[pz][e] * r_;
      }
    }
    if (s + 1 < T_) {                                  // first: the group waits for these
#pragma unroll
      for (int pz = 0; pz < 2; ++pz) {
        const int b = m0 + e_row + 8 * pz;
        if (b < a.B) {
          unsigned long long* xr = dx_w + (int64_t)b * gpr;
          Pair<T>::publish(xr, j, (unsigned)(s + 1), dr_pre[pz][0], dr_pre[pz][1]);
          Pair<T>::publish(xr, H + j, (unsigned)(s + 1), dz_pre[pz][0], dz_pre[pz][1]);
          Pair<T>::publish(xr, 2 * H + j, (unsigned)(s + 1), dnr[pz][0], dnr[pz][1]);
        }
      }
    }
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      const int b = m0 + e_row + 8 * pz;
      if (b < a.B) {
        const int64_t row = (int64_t)b * T_ + t;
        T* gi = (T*)a.dgi + row * a.ldgi + (int64_t)d * 3 * H + j;
        T* gh = (T*)a.dgh + row * a.ldgh + (int64_t)d * 3 * H + j;
        Pair<T>::st(gi, dr_pre[pz][0], dr_pre[pz][1]); Pair<T>::st(gi + H, dz_pre[pz][0], dz_pre[pz][1]); Pair<T>::st(gi + 2 * H, dn_pre[pz][0], dn_pre[pz][1]);
        Pair<T>::st(gh, dr_pre[pz][0], dr_pre[pz][1]); Pair<T>::st(gh + H, dz_pre[pz][0], dz_pre[pz][1]); Pair<T>::st(gh + 2 * H, dnr[pz][0], dnr[pz][1]);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Wide persistent BPTT (bf16, H = 32 * KS in {128, 256, 512}): the tiling and hand-off of gru_wide_fwd_kernel.  A group of H/64
// workgroups owns 16 rows of one direction; a wave owns 16 hidden units over the FULL K = 3H of dh_prev = dgh W_hh (its rows of
// W_hh^T: 3*KS fragments in registers, three accumulator chains by gate block); dgh of the previous BPTT step lives in an LDS
// image [16][3H] filled from the others' granules (3 * (KS/2 - 1) 16-byte sc1 loads per thread, requested right after this
// step's own hand-off) and from the own units.  Gate tape / dout / h_prev come in and dgi / dgh go out as staged 16-byte transfers.
// -DZS_GRU_PROFILE: one wave accumulates cycle counts per phase into the work header (tools/gru_bench.py, ZS_GRU_PROFILE_DUMP=1):
// of 13.3 k cycles per step 33 % are the sweep, 16 % the barrier, 32 % reading the image (all four waves read all 48 KB of it:
// 192 KB of LDS reads per step and CU) + MFMAs + W_hh^T fragments coming back from the AGPRs, 6 % gate math, 12 % hand-off + staging.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void gru_wide_bwd_kernel(const GruPersistBwdArgs a) {
  constexpr int H = 32 * KS;
  constexpr int NCH = 3 * (KS / 2 - 1);                 // sweep loads per thread: 16 rows x 3 gates x (H - 64) units / 4 / 256
  constexpr int NCHA = NCH > 0 ? NCH : 1;
  constexpr int CPR = (H - 64) / 4;                     // 16-byte chunks (4 units) of the other workgroups per (row, gate)
  constexpr int PITCH = 3 * H * 2 + 16;                 // bytes per row of the LDS image of dgh
  __shared__ __attribute__((aligned(16))) unsigned char gimg[2][16 * PITCH];       // dgh of the previous BPTT step, by step parity
  __shared__ __attribute__((aligned(16))) bf16_t in_s[2][6][16][64];                // r, z, n, hn, dout, h_prev of the step
  __shared__ __attribute__((aligned(16))) bf16_t out_s[2][4][16][64];               // dr_pre, dz_pre, dn_pre, dn_pre*r of the step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int d = blockIdx.z, T_ = a.T;
  const int m0 = blockIdx.y * 16;
  const int u_wg = blockIdx.x * 64;
  const int ul = wave * 16 + r;
  const int unit = u_wg + ul;
  constexpr int gpr = 3 * H / 2;                        // granules per row
  const int64_t par_stride = (int64_t)a.rows_pad * gpr;
  unsigned long long* dx_d = a.dx + (int64_t)d * 2 * par_stride;
  const gru_rsrc_t dx_rs = make_buffer_rsrc(dx_d, (unsigned)(2 * par_stride * 8));
  const unsigned par_bytes = (unsigned)(par_stride * 8);

  uint4 fb[3 * KS];                                      // rows (unit) of W_hh^T, K = 3H contiguous
  {
    const bf16_t* Wb = (const bf16_t*)a.whh_t + (int64_t)d * a.w_gstride + (int64_t)(u_wg + wave * 16 + r) * a.ldw + q * 8;
#pragma unroll
    for (int kk = 0; kk < 3 * KS; ++kk) fb[kk] = *reinterpret_cast<const uint4*>(Wb + kk * 32);
  }
  int ch_row[NCHA], ch_col[NCHA];
  unsigned ch_off[NCHA];
  bool ch_valid[NCHA];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + 256 * i;
    const int row = c / (3 * CPR), rem = c - row * (3 * CPR), g = rem / CPR, cu = (rem - g * CPR) * 4;
    ch_row[i] = row; ch_col[i] = g * H + (cu < u_wg ? cu : cu + 64);
    ch_valid[i] = (m0 + row) < a.B;
    ch_off[i] = (unsigned)(((int64_t)(m0 + row) * gpr + (ch_col[i] >> 1)) * 8);
  }
  // cooperative 16-byte transfers (8 units each): 6 input arrays x 16 rows x 8 chunks = 768 = 3 per thread; outputs likewise
  // (dgi: dr, dz, dn ; dgh: dr, dz, dn*r)
  auto in_src = [&](int id, int s_) -> const bf16_t* {
    const int arr = id >> 7, rem = id & 127, row = rem >> 3, c = rem & 7;
    const int b = min(m0 + row, a.B - 1);
    const int t = d == 0 ? T_ - 1 - s_ : s_;
    if (arr < 4) return (const bf16_t*)a.gates + (((int64_t)b * T_ + t) * 2 + d) * 4 * H + arr * H + u_wg + 8 * c;
    if (arr == 4) return (const bf16_t*)a.dout + ((int64_t)b * T_ + t) * a.ldd + a.dout_col + d * H + u_wg + 8 * c;
    const int tp = s_ < T_ - 1 ? (d == 0 ? t - 1 : t + 1) : t;            // last BPTT step: h_prev = 0 (zeroed below)
    return (const bf16_t*)a.out + ((int64_t)b * T_ + tp) * a.ldo + a.out_col + d * H + u_wg + 8 * c;
  };
  auto copy_out = [&](int par, int s_) {
    const int t = d == 0 ? T_ - 1 - s_ : s_;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int id = tid + 256 * k;
      const int arr = id >> 7, rem = id & 127, row = rem >> 3, c = rem & 7;     // arr 0..2: dgi r, z, n ; 3..5: dgh r, z, n
      const int b = m0 + row;
      if (b >= a.B) continue;
      const int g = arr < 3 ? arr : arr - 3;
      const int src = arr == 5 ? 3 : g;
      bf16_t* base = arr < 3 ? (bf16_t*)a.dgi + ((int64_t)b * T_ + t) * a.ldgi : (bf16_t*)a.dgh + ((int64_t)b * T_ + t) * a.ldgh;
      *reinterpret_cast<uint4*>(base + (int64_t)d * 3 * H + g * H + u_wg + 8 * c) = *reinterpret_cast<const uint4*>(&out_s[par][src][row][8 * c]);
    }
  };
  uint4 gq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) gq[k] = *reinterpret_cast<const uint4*>(in_src(tid + 256 * k, 0));
  float dhd[4] = {0.f, 0.f, 0.f, 0.f};                   // direct carry dh * z of the previous BPTT step
  bool dead = false;
#ifdef ZS_GRU_PROFILE
  long long pc[6] = {0, 0, 0, 0, 0, 0}, pt0 = 0, pt1 = 0;
#define ZS_PT(k) { pt1 = __builtin_readcyclecounter(); pc[k] += pt1 - pt0; pt0 = pt1; }
#else
#define ZS_PT(k)
#endif

  for (int s = 0; s < T_; ++s) {
#ifdef ZS_GRU_PROFILE
    pt0 = __builtin_readcyclecounter();
#endif
    unsigned char* const img = &gimg[s & 1][0];
    gu32x4_t g4[NCHA];
    f32x4_t acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if constexpr (NCH > 0) {
        // (requested here, not behind the previous hand-off as in the forward kernel: 21 x 4 registers held across the whole
        // step push the W_hh^T fragments through the AGPRs -- measured 7.6 against 7.05 us per step for the 4-wave K split)
        const unsigned poff = (unsigned)((s - 1) & 1) * par_bytes;
        unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int i = 0; i < NCH; ++i) g4[i] = load16_sc1(dx_rs, poff + ch_off[i]);
          bool ok = true;
#pragma unroll
          for (int i = 0; i < NCH; ++i) ok &= (((g4[i].y == (unsigned)s) & (g4[i].w == (unsigned)s)) | !ch_valid[i]);
          if (__all(ok) || dead) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > a.spin_limit) {
            if (lane == 0) {
              __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (a.status) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            dead = true;
            break;
          }
        }
        ZS_PT(0)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          *reinterpret_cast<uint2*>(img + ch_row[i] * PITCH + ch_col[i] * 2) = make_uint2(ch_valid[i] ? g4[i].x : 0u, ch_valid[i] ? g4[i].z : 0u);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int id = tid + 256 * k;
      uint4 v = gq[k];
      if ((id >> 7) == 5 && s == T_ - 1) v = make_uint4(0, 0, 0, 0);          // h_prev of the first time step is h0 = 0
      *reinterpret_cast<uint4*>(&in_s[s & 1][id >> 7][(id & 127) >> 3][8 * (id & 7)]) = v;
    }
    __syncthreads();                                     // images of dgh_{s-1} and of this step's operands are complete
    ZS_PT(1)
    if (s > 0) copy_out((s - 1) & 1, s - 1);
    if (s + 1 < T_) {
#pragma unroll
      for (int k = 0; k < 3; ++k) gq[k] = *reinterpret_cast<const uint4*>(in_src(tid + 256 * k, s + 1));
    }
    if (s > 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)                    // the three accumulator chains (gate blocks of K) interleaved
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const uint4 fa = *reinterpret_cast<const uint4*>(img + r * PITCH + q * 16 + (g * KS + ks) * 64);
          Frag16<bf16_t>::mma(fa, fb[g * KS + ks], acc[g]);
        }
    }
#ifdef ZS_GRU_PROFILE
    asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#endif
    ZS_PT(2)
    float dr_pre[4], dz_pre[4], dn_pre[4], dnr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * q + i;
      const float r_ = bf2f(in_s[s & 1][0][row][ul]), z_ = bf2f(in_s[s & 1][1][row][ul]), n_ = bf2f(in_s[s & 1][2][row][ul]);
      const float hn_ = bf2f(in_s[s & 1][3][row][ul]), do_ = bf2f(in_s[s & 1][4][row][ul]), hp_ = bf2f(in_s[s & 1][5][row][ul]);
      float dh = do_ + dhd[i];
      if (s > 0) dh += (acc[0][i] + acc[1][i]) + acc[2][i];
      const float dn = dh * (1.f - z_);
      const float dz = dh * (hp_ - n_);
      dhd[i] = dh * z_;
      dn_pre[i] = dn * (1.f - n_ * n_);
      dr_pre[i] = dn_pre[i] * hn_ * r_ * (1.f - r_);
      dz_pre[i] = dz * z_ * (1.f - z_);
      dnr[i] = dn_pre[i] * r_;
    }
    ZS_PT(3)
    if (s + 1 < T_) {                                    // hand-off first; even lanes publish (unit, unit + 1) of the three gate blocks
      unsigned long long* dx_w = dx_d + (int64_t)(s & 1) * par_stride;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float n0 = __shfl_xor(dr_pre[i], 1, 64), n1 = __shfl_xor(dz_pre[i], 1, 64), n2 = __shfl_xor(dnr[i], 1, 64);
        const int b = m0 + 4 * q + i;
        if (!(lane & 1) && b < a.B) {
          unsigned long long* xr = dx_w + (int64_t)b * gpr;
          store_granule(xr + (unit >> 1), (unsigned)(s + 1), Pair<bf16_t>::pack(dr_pre[i], n0));
          store_granule(xr + ((H + unit) >> 1), (unsigned)(s + 1), Pair<bf16_t>::pack(dz_pre[i], n1));
          store_granule(xr + ((2 * H + unit) >> 1), (unsigned)(s + 1), Pair<bf16_t>::pack(dnr[i], n2));
        }
      }
      unsigned char* const nimg = &gimg[(s + 1) & 1][0];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16_t* row = reinterpret_cast<bf16_t*>(nimg + (4 * q + i) * PITCH);
        row[unit] = f2bf(dr_pre[i]); row[H + unit] = f2bf(dz_pre[i]); row[2 * H + unit] = f2bf(dnr[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out_s[s & 1][0][4 * q + i][ul] = f2bf(dr_pre[i]); out_s[s & 1][1][4 * q + i][ul] = f2bf(dz_pre[i]);
      out_s[s & 1][2][4 * q + i][ul] = f2bf(dn_pre[i]); out_s[s & 1][3][4 * q + i][ul] = f2bf(dnr[i]);
    }
    ZS_PT(4)
  }
  __syncthreads();
  copy_out((T_ - 1) & 1, T_ - 1);
#ifdef ZS_GRU_PROFILE
  if (blockIdx.x == 2 && blockIdx.y == 3 && d == 0 && tid == 0) {
    long long* o = reinterpret_cast<long long*>(a.err) + 8;        // header bytes 64..
    for (int k = 0; k < 5; ++k) o[k] = pc[k];
  }
#endif
#undef ZS_PT
}

inline unsigned gate_blocks(int64_t total) {
  int64_t b = (total + NTG - 1) / NTG;
  if (b > 2048) b = 2048;
  return (unsigned)(b < 1 ? 1 : b);
}

// exchange granules of the persistent kernels: [2 dirs][2 parities][rows rounded to 32][cols * es / 4] x 8 bytes
// (cols = H forward, 3H backward)
size_t gru_hx_bytes(int B, int H, int es) {
  const size_t rows = (size_t)((B + 16 * RB - 1) / (16 * RB)) * 16 * RB;
  return (size_t)2 * 2 * rows * ((size_t)H * es / 4) * 8;
}
// options: relaxed atomics (zs_set_option may be called from any thread; a launch reads each knob once)
std::atomic<int> g_gru_persist{-1};
std::atomic<int> g_gru_wide{-1};
bool gru_wide_enabled() {
  int v = g_gru_wide.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("ZS_GRU_WIDE"); v = e ? atoi(e) : 1; g_gru_wide.store(v, std::memory_order_relaxed); }
  return v != 0;
}
std::atomic<int> g_gru_spin_limit{-1};
bool gru_persist_enabled() {
  int v = g_gru_persist.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("ZS_GRU_PERSIST"); v = e ? atoi(e) : 1; g_gru_persist.store(v, std::memory_order_relaxed); }
  return v != 0;
}
unsigned gru_spin_limit() {
  int v = g_gru_spin_limit.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("ZS_GRU_SPIN_LIMIT"); v = e ? atoi(e) : (int)GRU_SPIN_LIMIT; g_gru_spin_limit.store(v, std::memory_order_relaxed); }
  return (unsigned)v;
}
// workgroups that are certainly co-resident: one per CU
int64_t gru_resident_limit() {
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_relaxed);
  if (c == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
    if (c <= 0) c = 256;
    cus.store(c, std::memory_order_relaxed);
  }
  return c;
}

}  // namespace

// "gru_wide" knob of zs_set_option: the 16-row x 64-unit tiling of the persistent kernels (bf16, H in {128, 256, 512})
int zs_gru_wide_option(int value) { const int old = gru_wide_enabled() ? 1 : 0; g_gru_wide.store(value ? 1 : 0, std::memory_order_relaxed); return old; }
// "gru_persist" knob of zs_set_option
int zs_gru_persist_option(int value) { const int old = gru_persist_enabled() ? 1 : 0; g_gru_persist.store(value ? 1 : 0, std::memory_order_relaxed); return old; }
// "gru_spin_limit": sweeps a persistent GRU wave waits for its group before it gives up (tests force a timeout with 0)
int zs_gru_spin_limit_option(int value) { const int old = (int)gru_spin_limit(); g_gru_spin_limit.store(value < 0 ? 0 : value, std::memory_order_relaxed); return old; }

extern "C" size_t zs_gru_work_bytes(int32_t B, int32_t H) {
  // gh [2][B][3H] + hstate/dhd [2][B][H] + dhg [2][B][H], fp32; or the persistent kernel's exchange granules (fp32 size) + error word
  const size_t steps = (size_t)2 * (size_t)B * (size_t)(5 * H) * sizeof(float) + 256;
  const size_t hx = gru_hx_bytes(B, 3 * H, 4) + 256;
  return steps > hx ? steps : hx;
}

// Debug / test hook: read back the error word of the persistent kernels (set when a bounded spin timed out; the first word of the
// work buffer's header, which the one-launch-per-step path never touches).  Synchronises.
extern "C" int zs_gru_check(const float* work, int32_t B, int32_t H, void* stream) {
  ZS_REQUIRE(work && B > 0 && H > 0, "zs_gru_check: bad args");
  unsigned v = 0;
  if (hipMemcpyAsync(&v, reinterpret_cast<const char*>(work), sizeof(v), hipMemcpyDeviceToHost,
                     (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    zs_set_error("zs_gru_check: copy failed");
    return ZS_ELAUNCH;
  }
  if (v != 0) { zs_set_error("zs_gru_check: a persistent GRU workgroup timed out waiting for its group (results are invalid)"); return ZS_ELAUNCH; }
  return ZS_OK;
}

// the wide kernels move gate inputs / outputs as 16-byte vectors
static bool gru_wide_aligned(const ZsGruFwd* p) {
  return (p->out_col % 8) == 0 && (p->ldo % 8) == 0 && (p->ldgi % 8) == 0 && ((uintptr_t)p->out % 16) == 0 && ((uintptr_t)p->gi % 16) == 0 &&
         (!p->gates || ((uintptr_t)p->gates % 16) == 0) && (!p->bcast_vec || (p->bcast_col % 8) == 0);
}

extern "C" int zs_gru_fwd(const ZsGruFwd* p, void* stream) {
  ZS_REQUIRE(p && p->gi && p->whh && p->bhh && p->out && p->work, "zs_gru_fwd: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_gru_fwd: bad dtype");
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->H > 0 && p->H % 8 == 0, "zs_gru_fwd: sizes (H %% 8 == 0 required, H=%d)", p->H);
  ZS_REQUIRE(p->work_bytes >= zs_gru_work_bytes(p->B, p->H), "zs_gru_fwd: work buffer too small");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  const int kc = 128 / es;
  ZS_REQUIRE((p->out_col * es) % 16 == 0, "zs_gru_fwd: out_col alignment");
  const int B = p->B, T = p->T, H = p->H;
  float* const wbase = p->work + GRU_WORK_HDR / sizeof(float);
  float* gh = wbase;
  float* hstate = wbase + (size_t)2 * B * 3 * H;
  const char* outb = (const char*)p->out;
  ZS_REQUIRE(!p->bcast_vec || (p->bcast_idx && (p->bcast_col * es) % 4 == 0), "zs_gru_fwd: bcast_idx / bcast_col");
  bool bcast_pending = p->bcast_vec != nullptr;        // every path but the persistent kernel: a launch of its own, first
  {
    const int nrb0 = (B + 16 * RB - 1) / (16 * RB);
    const int kstep0 = p->dtype == ZS_F32 ? 16 : 32;
    const int per0 = (H % (4 * kstep0) == 0) ? H / (4 * kstep0) : 0;
    if (p->whh_interleaved && H % 32 == 0 && gru_persist_enabled() && (per0 == 1 || per0 == 2 || per0 == 4) &&
        (int64_t)(H / 32) * nrb0 * 2 <= gru_resident_limit() && T > 1)
      bcast_pending = false;
    if (p->whh_interleaved && gru_persist_enabled() && gru_wide_enabled() && p->dtype == ZS_BF16 && (H == 128 || H == 256 || H == 512) && T > 1 &&
        (int64_t)(H / 64) * ((B + 15) / 16) * 2 <= gru_resident_limit() && gru_wide_aligned(p))
      bcast_pending = false;
  }
  if (bcast_pending) {
    ZsAddRowvec r;
    memset(&r, 0, sizeof(r));
    r.dtype = p->dtype; r.vec = p->bcast_vec; r.vec_ld = p->bcast_ld; r.idx = p->bcast_idx;
    r.out = (void*)(outb + (int64_t)p->bcast_col * es); r.ldo = p->ldo; r.B = B; r.T = T; r.C = 2 * H; r.fill_cols = 2 * H;
    int rc = zs_add_rowvec(&r, stream);
    if (rc) return rc;
  }
  if (p->whh_interleaved && H % 32 == 0) {
    // persistent kernel: the time loop runs on the device (see gru_persist_fwd_kernel)
    const int nrb = (B + 16 * RB - 1) / (16 * RB);
    const int kstep = p->dtype == ZS_F32 ? 16 : 32;
    const int per = (H % (4 * kstep) == 0) ? H / (4 * kstep) : 0;
    const int64_t nwg = (int64_t)(H / 32) * nrb * 2;
    const int nrb16 = (B + 15) / 16;
    const bool wide = gru_persist_enabled() && gru_wide_enabled() && p->dtype == ZS_BF16 && (H == 128 || H == 256 || H == 512) && T > 1 &&
                      (int64_t)(H / 64) * nrb16 * 2 <= gru_resident_limit() && gru_wide_aligned(p);
    if (wide || (gru_persist_enabled() && (per == 1 || per == 2 || per == 4) && nwg <= gru_resident_limit() && T > 1)) {
      GruPersistArgs a;
      memset(&a, 0, sizeof(a));
      a.gi = p->gi; a.ldgi = p->ldgi; a.whh = p->whh; a.ldw = p->ldw; a.w_gstride = p->w_gstride;
      a.bhh = p->bhh; a.bhh_gstride = p->bhh_gstride; a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col; a.gates = p->gates;
      const size_t hx_bytes = gru_hx_bytes(B, H, es);
      a.hx = reinterpret_cast<unsigned long long*>(wbase);
      a.err = reinterpret_cast<unsigned*>(p->work);
      a.status = p->status; a.spin_limit = gru_spin_limit();
      a.B = B; a.T = T; a.H = H; a.rows_pad = wide ? nrb16 * 16 : nrb * 16 * RB;
      a.bcast_vec = p->bcast_vec; a.bcast_ld = p->bcast_ld; a.bcast_idx = p->bcast_idx; a.bcast_col = p->bcast_col;
      a.dir1_forward = p->dir1_forward != 0;
      if (hipMemsetAsync(p->work, 0, GRU_WORK_HDR + hx_bytes, (hipStream_t)stream) != hipSuccess) {
        zs_set_error("zs_gru_fwd: memset failed");
        return ZS_ELAUNCH;
      }
      if (wide) {
        dim3 wgrid(H / 64, nrb16, 2);
        if (H == 512) hipLaunchKernelGGL(gru_wide_fwd_kernel<16>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
        else if (H == 256) hipLaunchKernelGGL(gru_wide_fwd_kernel<8>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(gru_wide_fwd_kernel<4>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
        return zs_check_launch("zs_gru_fwd.wide");
      }
      dim3 grid(H / 32, nrb, 2);
#define ZS_GRU_PF(TT, PP) hipLaunchKernelGGL((gru_persist_fwd_kernel<TT, PP>), grid, dim3(256), 0, (hipStream_t)stream, a)
      if (p->dtype == ZS_F32) { if (per == 1) ZS_GRU_PF(float, 1); else if (per == 2) ZS_GRU_PF(float, 2); else ZS_GRU_PF(float, 4); }
      else { if (per == 1) ZS_GRU_PF(bf16_t, 1); else if (per == 2) ZS_GRU_PF(bf16_t, 2); else ZS_GRU_PF(bf16_t, 4); }
#undef ZS_GRU_PF
      return zs_check_launch("zs_gru_fwd.persist");
    }
    // one fused launch per step: recurrent product + gates (see rowblock_gemm_kernel)
    for (int s = 0; s < T; ++s) {
      RowGemmArgs a;
      memset(&a, 0, sizeof(a));
      const int sp = s > 0 ? s - 1 : 0;
      const int64_t off0 = (int64_t)sp * p->ldo + p->out_col;                        // dir 0 reads h_{t-1} at row t-1
      const int64_t off1 = (int64_t)(p->dir1_forward ? sp : (s > 0 ? T - s : T - 1)) * p->ldo + p->out_col + H;   // dir 1 reads row t+1 (t-1 when it runs forward)
      a.A = outb + off0 * es; a.a_row_stride = (int64_t)T * p->ldo; a.a_gstride = off1 - off0;
      a.W = p->whh; a.ldw = p->ldw; a.w_gstride = p->w_gstride;
      a.M = B; a.N = 3 * H; a.K = H;
      GateFwdArgs& g = a.gate;
      g.gi = p->gi; g.ldgi = p->ldgi; g.gh = nullptr; g.bhh = p->bhh; g.bhh_gstride = p->bhh_gstride; g.hstate = hstate;
      g.out = p->out; g.ldo = p->ldo; g.out_col = p->out_col; g.gates = p->gates; g.B = B; g.T = T; g.H = H; g.step = s;
      g.dir1_forward = p->dir1_forward != 0;
      dim3 grid(H / 32, (B + 16 * RB - 1) / (16 * RB), 2);
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((rowblock_gemm_kernel<float, 6, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((rowblock_gemm_kernel<bf16_t, 6, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
      int rc = zs_check_launch("zs_gru_fwd.step");
      if (rc) return rc;
    }
    return ZS_OK;
  }
  for (int s = 0; s < T; ++s) {
    if (s > 0) {
      ZsGemmConv g;
      memset(&g, 0, sizeof(g));
      g.dtype = p->dtype;
      const int64_t off0 = (int64_t)(s - 1) * p->ldo + p->out_col;
      const int64_t off1 = (int64_t)(p->dir1_forward ? s - 1 : T - s) * p->ldo + p->out_col + H;
      g.A = outb + off0 * es; g.lda = p->ldo; g.a_batch_stride = (int64_t)T * p->ldo;
      g.a_gstride = off1 - off0;
      g.B = B; g.T_in = 1; g.T_out = 1; g.taps = 1; g.stride = 1; g.pad_left = 0; g.pad_mode = ZS_PAD_ZERO; g.gather = 0;
      g.cin_pad = ((H + kc - 1) / kc) * kc;
      g.W = p->whh; g.ldw = p->ldw; g.w_gstride = p->w_gstride; g.N = 3 * H; g.n_pad = p->n_pad;
      g.act = ZS_ACT_NONE;
      g.out = gh; g.ldc = 3 * H; g.out_f32 = 1; g.out_cols = 3 * H; g.store_mode = ZS_STORE_ROWS; g.out_gstride = (int64_t)B * 3 * H;
      g.groups = 2;
      int rc = zs_gemm_conv(&g, stream);
      if (rc) return rc;
    }
    GateFwdArgs a;
    a.gi = p->gi; a.ldgi = p->ldgi; a.gh = gh; a.bhh = p->bhh; a.bhh_gstride = p->bhh_gstride; a.hstate = hstate;
    a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col; a.gates = p->gates; a.B = B; a.T = T; a.H = H; a.step = s;
    a.dir1_forward = p->dir1_forward != 0;
    const unsigned nb = gate_blocks((int64_t)2 * B * H);
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(gru_gate_fwd_kernel<float>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gru_gate_fwd_kernel<bf16_t>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
    int rc = zs_check_launch("zs_gru_fwd.gate");
    if (rc) return rc;
  }
  return ZS_OK;
}

extern "C" int zs_gru_bwd(const ZsGruBwd* p, void* stream) {
  ZS_REQUIRE(p && p->dout && p->out && p->gates && p->whh_t && p->dgi && p->dgh && p->work, "zs_gru_bwd: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_gru_bwd: bad dtype");
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->H > 0 && p->H % 8 == 0, "zs_gru_bwd: sizes");
  ZS_REQUIRE(p->work_bytes >= zs_gru_work_bytes(p->B, p->H), "zs_gru_bwd: work buffer too small");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  const int kc = 128 / es;
  const int B = p->B, T = p->T, H = p->H;
  float* const wbase = p->work + GRU_WORK_HDR / sizeof(float);
  float* dhd = wbase + (size_t)2 * B * 3 * H;
  float* dhg = dhd + (size_t)2 * B * H;
  const char* dghb = (const char*)p->dgh;
  const bool fast = (H % 32 == 0);
  if (fast && gru_persist_enabled() && T > 1) {
    const int nrb = (B + 15) / 16;                         // 16 rows x 64 units per workgroup
    if (gru_wide_enabled() && p->dtype == ZS_BF16 && (H == 128 || H == 256 || H == 512) && (int64_t)(H / 64) * nrb * 2 <= gru_resident_limit() &&
        (p->dout_col % 8) == 0 && (p->out_col % 8) == 0 && (p->ldd % 8) == 0 && (p->ldo % 8) == 0 && (p->ldgi % 8) == 0 && (p->ldgh % 8) == 0 &&
        (((uintptr_t)p->dout | (uintptr_t)p->out | (uintptr_t)p->gates | (uintptr_t)p->dgi | (uintptr_t)p->dgh) % 16) == 0) {
      GruPersistBwdArgs a;
      memset(&a, 0, sizeof(a));
      a.dout = p->dout; a.ldd = p->ldd; a.dout_col = p->dout_col; a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col;
      a.gates = p->gates; a.whh_t = p->whh_t; a.ldw = p->ldw; a.w_gstride = p->w_gstride;
      a.dgi = p->dgi; a.ldgi = p->ldgi; a.dgh = p->dgh; a.ldgh = p->ldgh;
      const size_t dx_bytes = gru_hx_bytes(B, 3 * H, es);
      a.dx = reinterpret_cast<unsigned long long*>(wbase);
      a.err = reinterpret_cast<unsigned*>(p->work);
      a.status = p->status; a.spin_limit = gru_spin_limit();
      a.B = B; a.T = T; a.H = H; a.rows_pad = nrb * 16;
      if (hipMemsetAsync(p->work, 0, GRU_WORK_HDR + dx_bytes, (hipStream_t)stream) != hipSuccess) {
        zs_set_error("zs_gru_bwd: memset failed");
        return ZS_ELAUNCH;
      }
      dim3 wgrid(H / 64, nrb, 2);
      if (H == 512) hipLaunchKernelGGL(gru_wide_bwd_kernel<16>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
      else if (H == 256) hipLaunchKernelGGL(gru_wide_bwd_kernel<8>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(gru_wide_bwd_kernel<4>, wgrid, dim3(256), 0, (hipStream_t)stream, a);
      return zs_check_launch("zs_gru_bwd.wide");
    }
    const int kstep = p->dtype == ZS_F32 ? 16 : 32;
    const int perb = ((3 * H) % (4 * kstep) == 0 && H % (16 * BNT) == 0) ? (3 * H) / (4 * kstep) : 0;
    const int64_t nwg = (int64_t)(H / (16 * BNT)) * nrb * 2;
    if ((perb == 3 || perb == 6 || perb == 12) && nwg <= gru_resident_limit()) {
      GruPersistBwdArgs a;
      memset(&a, 0, sizeof(a));
      a.dout = p->dout; a.ldd = p->ldd; a.dout_col = p->dout_col; a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col;
      a.gates = p->gates; a.whh_t = p->whh_t; a.ldw = p->ldw; a.w_gstride = p->w_gstride;
      a.dgi = p->dgi; a.ldgi = p->ldgi; a.dgh = p->dgh; a.ldgh = p->ldgh;
      const size_t dx_bytes = gru_hx_bytes(B, 3 * H, es);
      a.dx = reinterpret_cast<unsigned long long*>(wbase);
      a.err = reinterpret_cast<unsigned*>(p->work);
      a.status = p->status; a.spin_limit = gru_spin_limit();
      a.B = B; a.T = T; a.H = H; a.rows_pad = nrb * 16;
      if (hipMemsetAsync(p->work, 0, GRU_WORK_HDR + dx_bytes, (hipStream_t)stream) != hipSuccess) {
        zs_set_error("zs_gru_bwd: memset failed");
        return ZS_ELAUNCH;
      }
      dim3 grid(H / (16 * BNT), nrb, 2);
#define ZS_GRU_PB(TT, PP) hipLaunchKernelGGL((gru_persist_bwd_kernel<TT, PP>), grid, dim3(256), 0, (hipStream_t)stream, a)
      if (p->dtype == ZS_F32) { if (perb == 3) ZS_GRU_PB(float, 3); else if (perb == 6) ZS_GRU_PB(float, 6); else ZS_GRU_PB(float, 12); }
      else { if (perb == 3) ZS_GRU_PB(bf16_t, 3); else if (perb == 6) ZS_GRU_PB(bf16_t, 6); else ZS_GRU_PB(bf16_t, 12); }
#undef ZS_GRU_PB
      return zs_check_launch("zs_gru_bwd.persist");
    }
  }
  for (int s = 0; s < T; ++s) {
    GateBwdArgs a;
    a.dout = p->dout; a.ldd = p->ldd; a.dout_col = p->dout_col; a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col;
    a.gates = p->gates; a.dhd = dhd; a.dhg = dhg; a.dgi = p->dgi; a.ldgi = p->ldgi; a.dgh = p->dgh; a.ldgh = p->ldgh;
    a.B = B; a.T = T; a.H = H; a.step = s;
    int rc = ZS_OK;
    if (s == 0 || !fast) {                       // fast path: steps >= 1 have their gate math fused into the product below
      const unsigned nb = gate_blocks((int64_t)2 * B * H);
      if (p->dtype == ZS_F32) hipLaunchKernelGGL(gru_gate_bwd_kernel<float>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(gru_gate_bwd_kernel<bf16_t>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
      rc = zs_check_launch("zs_gru_bwd.gate");
      if (rc) return rc;
    }
    if (s < T - 1 && fast) {
      // dh carry of step s+1 = dgh_s W_hh (MFMA product) ; epilogue = gate backward of step s+1
      RowGemmArgs ra;
      memset(&ra, 0, sizeof(ra));
      const int64_t off0 = (int64_t)(T - 1 - s) * p->ldgh;
      const int64_t off1 = (int64_t)s * p->ldgh + 3 * H;
      ra.A = dghb + off0 * es; ra.a_row_stride = (int64_t)T * p->ldgh; ra.a_gstride = off1 - off0;
      ra.W = p->whh_t; ra.ldw = p->ldw; ra.w_gstride = p->w_gstride;
      ra.M = B; ra.N = H; ra.K = 3 * H;
      ra.gbw = a; ra.gbw.step = s + 1;
      dim3 grid(H / 32, (B + 16 * RB - 1) / (16 * RB), 2);
      if (p->dtype == ZS_F32) hipLaunchKernelGGL((rowblock_gemm_kernel<float, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, ra);
      else hipLaunchKernelGGL((rowblock_gemm_kernel<bf16_t, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, ra);
      rc = zs_check_launch("zs_gru_bwd.step");
      if (rc) return rc;
    } else if (s < T - 1) {
      ZsGemmConv g;
      memset(&g, 0, sizeof(g));
      g.dtype = p->dtype;
      const int64_t off0 = (int64_t)(T - 1 - s) * p->ldgh;            // dir 0 row t = T-1-s, columns [0,3H)
      const int64_t off1 = (int64_t)s * p->ldgh + 3 * H;              // dir 1 row t = s,     columns [3H,6H)
      g.A = dghb + off0 * es; g.lda = p->ldgh; g.a_batch_stride = (int64_t)T * p->ldgh; g.a_gstride = off1 - off0;
      g.B = B; g.T_in = 1; g.T_out = 1; g.taps = 1; g.stride = 1; g.pad_left = 0; g.pad_mode = ZS_PAD_ZERO; g.gather = 0;
      g.cin_pad = ((3 * H + kc - 1) / kc) * kc;
      g.W = p->whh_t; g.ldw = p->ldw; g.w_gstride = p->w_gstride; g.N = H; g.n_pad = p->n_pad;
      g.act = ZS_ACT_NONE;
      g.out = dhg; g.ldc = H; g.out_f32 = 1; g.out_cols = H; g.store_mode = ZS_STORE_ROWS; g.out_gstride = (int64_t)B * H;
      g.groups = 2;
      rc = zs_gemm_conv(&g, stream);
      if (rc) return rc;
    }
  }
  return ZS_OK;
}
