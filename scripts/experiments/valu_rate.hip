// valu_rate.hip - how many cycles does one wave64 VALU instruction cost a gfx950 SIMD?
//   plain v_fma_f32 vs packed v_pk_fma_f32, at 1 / 2 / 4 wavefronts per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int PACKED> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  if constexpr (PACKED == 0) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f2 acc[16];
    f2 av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int iters = 20000;
  for (int packed = 0; packed < 2; ++packed)
    for (int wps : {1, 2, 4, 8}) {
      dim3 grid(256 * wps), block(256);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (packed) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      double instr_per_simd = (double)iters * 64 * wps; // wave-instructions issued on one SIMD
      double cycles = ms * 1e-3 * clk_khz * 1e3;
      printf("%s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (clock %d MHz), %.1f TFLOP/s\n",
             packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, cycles / instr_per_simd, clk_khz / 1000,
             instr_per_simd * 1024 * 64 * (packed ? 4 : 2) / (ms * 1e-3) / 1e12);
    }
  return 0;
}
