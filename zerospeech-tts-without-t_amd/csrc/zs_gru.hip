// zs_gru.hip -- bidirectional GRU (model/model.py:59-66; nn.GRU gate order r,z,n, zero h0).
// The input projections for all T and both directions are one zs_gemm_conv call made by the caller.
// Here: the sequential part.  Per time step, ONE grouped MFMA product (both directions as groups,
// h_{t-1} W_hh^T, fp32 out) and ONE gate kernel are enqueued; the loop over T lives in the library so
// the host pays one C call per layer, and the whole layer is graph-capturable.
//   forward :  r = s(gi_r + gh_r + b_hr)  z = s(gi_z + gh_z + b_hz)  n = tanh(gi_n + r*(gh_n + b_hn))
//              h = (1-z)*n + z*h_prev
//   backward:  BPTT with dh carried as (direct part dh*z) + (dgh W_hh) computed by the same product kernel.
#include "zs_common.h"

namespace {

constexpr int NTG = 256;

struct GateFwdArgs {
  const void* gi; int64_t ldgi;
  const float* gh;        // [2][B][3H]
  const float* bhh; int64_t bhh_gstride;
  float* hstate;          // [2][B][H]
  void* out; int64_t ldo; int out_col;
  void* gates;            // [B][T][2][4H] or null
  int B, T, H, step;
};

template <typename T>
__global__ void gru_gate_fwd_kernel(const GateFwdArgs a) {
  const int64_t total = (int64_t)2 * a.B * a.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / ((int64_t)a.B * a.H));
    const int64_t rem = i - (int64_t)d * a.B * a.H;
    const int b = (int)(rem / a.H), j = (int)(rem - (int64_t)b * a.H);
    const int t = d == 0 ? a.step : a.T - 1 - a.step;
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

inline unsigned gate_blocks(int64_t total) {
  int64_t b = (total + NTG - 1) / NTG;
  if (b > 2048) b = 2048;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" size_t zs_gru_work_bytes(int32_t B, int32_t H) {
  // gh [2][B][3H] + hstate/dhd [2][B][H] + dhg [2][B][H], fp32
  return (size_t)2 * (size_t)B * (size_t)(5 * H) * sizeof(float) + 256;
}

extern "C" int zs_gru_fwd(const ZsGruFwd* p, void* stream) {
  ZS_REQUIRE(p && p->gi && p->whh && p->bhh && p->out && p->work, "zs_gru_fwd: null operand");
  ZS_REQUIRE(p->dtype == ZS_F32 || p->dtype == ZS_BF16, "zs_gru_fwd: bad dtype");
  ZS_REQUIRE(p->B > 0 && p->T > 0 && p->H > 0 && p->H % 8 == 0, "zs_gru_fwd: sizes (H %% 8 == 0 required, H=%d)", p->H);
  ZS_REQUIRE(p->work_bytes >= zs_gru_work_bytes(p->B, p->H), "zs_gru_fwd: work buffer too small");
  const int es = p->dtype == ZS_F32 ? 4 : 2;
  ZS_REQUIRE((p->out_col * es) % 16 == 0, "zs_gru_fwd: out_col alignment");
  const int B = p->B, T = p->T, H = p->H;
  float* gh = p->work;
  float* hstate = p->work + (size_t)2 * B * 3 * H;
  const char* outb = (const char*)p->out;
  for (int s = 0; s < T; ++s) {
    if (s > 0) {
      ZsGemmConv g;
      memset(&g, 0, sizeof(g));
      g.dtype = p->dtype;
      const int64_t off0 = (int64_t)(s - 1) * p->ldo + p->out_col;
      const int64_t off1 = (int64_t)(T - s) * p->ldo + p->out_col + H;
      g.A = outb + off0 * es; g.lda = p->ldo; g.a_batch_stride = (int64_t)T * p->ldo;
      g.a_gstride = off1 - off0;
      g.B = B; g.T_in = 1; g.T_out = 1; g.taps = 1; g.stride = 1; g.pad_left = 0; g.pad_mode = ZS_PAD_ZERO; g.gather = 0;
      g.cin_pad = ((H + 31) / 32) * 32;
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
  const int B = p->B, T = p->T, H = p->H;
  float* dhd = p->work + (size_t)2 * B * 3 * H;
  float* dhg = dhd + (size_t)2 * B * H;
  const char* dghb = (const char*)p->dgh;
  for (int s = 0; s < T; ++s) {
    GateBwdArgs a;
    a.dout = p->dout; a.ldd = p->ldd; a.dout_col = p->dout_col; a.out = p->out; a.ldo = p->ldo; a.out_col = p->out_col;
    a.gates = p->gates; a.dhd = dhd; a.dhg = dhg; a.dgi = p->dgi; a.ldgi = p->ldgi; a.dgh = p->dgh; a.ldgh = p->ldgh;
    a.B = B; a.T = T; a.H = H; a.step = s;
    const unsigned nb = gate_blocks((int64_t)2 * B * H);
    if (p->dtype == ZS_F32) hipLaunchKernelGGL(gru_gate_bwd_kernel<float>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gru_gate_bwd_kernel<bf16_t>, dim3(nb), dim3(NTG), 0, (hipStream_t)stream, a);
    int rc = zs_check_launch("zs_gru_bwd.gate");
    if (rc) return rc;
    if (s < T - 1) {
      ZsGemmConv g;
      memset(&g, 0, sizeof(g));
      g.dtype = p->dtype;
      const int64_t off0 = (int64_t)(T - 1 - s) * p->ldgh;            // dir 0 row t = T-1-s, columns [0,3H)
      const int64_t off1 = (int64_t)s * p->ldgh + 3 * H;              // dir 1 row t = s,     columns [3H,6H)
      g.A = dghb + off0 * es; g.lda = p->ldgh; g.a_batch_stride = (int64_t)T * p->ldgh; g.a_gstride = off1 - off0;
      g.B = B; g.T_in = 1; g.T_out = 1; g.taps = 1; g.stride = 1; g.pad_left = 0; g.pad_mode = ZS_PAD_ZERO; g.gather = 0;
      g.cin_pad = ((3 * H + 31) / 32) * 32;
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
