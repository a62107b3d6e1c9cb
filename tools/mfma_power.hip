// Pure-MFMA power wall probe: each wave loops over register-resident operands (no LDS, no memory) with either
// v_mfma_f32_32x32x16_bf16 or v_mfma_f32_16x16x32_bf16, operands random or zero.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xffff]; b[i] = in[(tid * 8 + 4 + i) & 0xffff]; }
  if (KIND == 0) {
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + j) & 3]), c[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    out[tid] = s;
  } else {
    f32x4 c[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + j) & 3]), c[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += c[j][r];
    out[tid] = s;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint4* in; float* out;
  hipMalloc(&in, 65536 * 16); hipMalloc(&out, 256 * 2 * 512 * 4);
  for (int fill = 0; fill < 3; ++fill) {
    std::vector<unsigned short> h(65536 * 8);
    for (auto& v : h) {
      if (fill == 0) v = 0;
      else if (fill == 1) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
      else { float f = 0; for (int q = 0; q < 12; ++q) f += rand() / (float)RAND_MAX; f -= 6; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 2; ++kind) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256 * 2), dim3(512), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(256 * 2), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (kind == 0 ? 16.0 * 32768 : 32.0 * 16384) * (double)iters * 512 * 8;
      printf("%s fill=%s: %.2f ms  %.0f TFLOP/s\n", kind == 0 ? "32x32x16" : "16x16x32", fill == 0 ? "zeros" : fill == 1 ? "uniform" : "normal", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
