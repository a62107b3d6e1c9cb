// zs_common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/zs_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;   // raw bf16 bits in memory

void zs_set_error(const char* fmt, ...);
int zs_check_launch(const char* what);
int zs_gl_prefetch_option(int value);    // zs_griffin.hip: "gl_prefetch" knob
int zs_norm_wide_option(int value);      // zs_norm.hip: "norm_wide" knob (8 row groups x 256 channels for T <= 64)
int zs_norm_lim_option(int i, int value);
int zs_gl_chains_option(int value);      // zs_griffin.hip: "gl_chains" knob (independent launch chains of zs_griffin_lim)
int zs_gru_spin_limit_option(int value); // zs_gru.hip: "gru_spin_limit" knob, returns the previous value
int zs_gru_persist_option(int value);   // zs_gru.hip: "gru_persist" knob, returns the previous value
int zs_gru_wide_option(int value);      // zs_gru.hip: "gru_wide" knob, returns the previous value

#define ZS_REQUIRE(cond, ...)                  \
  do {                                         \
    if (!(cond)) {                             \
      zs_set_error(__VA_ARGS__);               \
      return ZS_EINVAL;                        \
    }                                          \
  } while (0)

// ---- element conversion ------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // round-to-nearest-even, NaN preserved via the compiler's cast (v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPer16B = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 8 consecutive elements <-> 8 floats (one 16-B access for bf16, two for fp32). p must be 16-B aligned.
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  uint4 a = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
// the same 8 elements kept raw (as loaded) so that a prefetch does not park the wave on the load latency at the load site
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
  __device__ __forceinline__ void cvt(float (&v)[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw8<bf16_t> {
  uint4 a;
  __device__ __forceinline__ void ld(const bf16_t* p) { a = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { a = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void cvt(float (&v)[8]) const {
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};

template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- counter-hash RNG (dropout keep mask, Gumbel uniforms) ----------------------------------------
__device__ __forceinline__ uint32_t zs_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t zs_rand32(uint64_t seed, uint32_t stream_id, uint64_t idx) {
  uint32_t a = zs_mix32((uint32_t)idx ^ (uint32_t)seed);
  uint32_t b = zs_mix32((uint32_t)(idx >> 32) + (uint32_t)(seed >> 32) + stream_id * 0x9e3779b9U + a);
  return zs_mix32(a ^ (b * 0x85ebca6bU) ^ stream_id);
}
__device__ __forceinline__ float zs_uniform(uint64_t seed, uint32_t stream_id, uint64_t idx) {
  return (float)(zs_rand32(seed, stream_id, idx) >> 8) * (1.0f / 16777216.0f);   // [0,1), 24 bits
}
__device__ __forceinline__ bool zs_keep(uint64_t seed, uint32_t stream_id, uint64_t idx, float p) {
  return zs_uniform(seed, stream_id, idx) >= p;
}

// ---- reflect index (F.pad mode='reflect'): requires pad < T ------------------------------------
__device__ __host__ __forceinline__ int zs_reflect(int s, int T) {
  if (s < 0) s = -s;
  if (s >= T) s = 2 * (T - 1) - s;
  return s;
}

// ---- wave / block reductions -----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float dlrelu_f(float y, float slope) { return y > 0.f ? 1.f : slope; }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + __expf(-v)); }
