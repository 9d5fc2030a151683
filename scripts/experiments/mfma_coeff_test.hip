// Stand-alone check of the MFMA bicubic-coefficient build (exactness + operand layouts).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_coeff_test.hip -o mfma_coeff_test && ./mfma_coeff_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void cubic_1d(float p0, float p1, float p2, float p3, float &c0, float &c1, float &c2,
                                         float &c3) {
  c0 = __builtin_fmaf(2.0f, p0, __builtin_fmaf(-3.0f, p1, __builtin_fmaf(3.0f, p2, -p3)));
  c1 = __builtin_fmaf(-4.0f, p0, __builtin_fmaf(9.5f, p1, __builtin_fmaf(-8.0f, p2, 2.5f * p3)));
  c2 = __builtin_fmaf(2.5f, p0, __builtin_fmaf(-7.0f, p1, __builtin_fmaf(6.5f, p2, -2.0f * p3)));
  c3 = __builtin_fmaf(-0.5f, p0, __builtin_fmaf(1.5f, p1, __builtin_fmaf(-1.5f, p2, 0.5f * p3)));
}

__device__ __forceinline__ float cm(int a, int b) {
  const float C[4][4] = {{2.0f, -3.0f, 3.0f, -1.0f}, {-4.0f, 9.5f, -8.0f, 2.5f}, {2.5f, -7.0f, 6.5f, -2.0f},
                         {-0.5f, 1.5f, -1.5f, 0.5f}};
  return C[a][b];
}

// A operand of MFMA #target: coefficient c sits at output row (c&3) + 8*(c>>2) + 4*target
__device__ half8 make_A(int lane, int target) {
  const int i = lane & 31, hk = lane >> 5;
  half8 a;
  const int ip = i - 4 * target;
  const bool used = ip >= 0 && (ip & 4) == 0;
  const int c = (ip & 3) + 4 * (ip >> 3);
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * hk + j; // pixel index 4*r + col
    float v = used ? cm(c >> 2, k >> 2) * cm(c & 3, k & 3) : 0.f;
    a[j] = (_Float16)v;
  }
  return a;
}

// two bytes of a dword -> packed (1024 + b_lo, 1024 + b_hi) as f16x2
__device__ __forceinline__ uint32_t bytes_to_h2(uint32_t w, bool hi) {
  // v_perm_b32 D = {S0, S1} bytes; selector picks: byte index 0-3 from S1, 4-7 from S0
  return hi ? __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u) : __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u);
}

__global__ void k(const uint8_t *win /*[64][16]*/, float *ref /*[64][16]*/, float *got /*[64][16]*/) {
  const int lane = threadIdx.x;
  uint32_t r[4];
  for (int q = 0; q < 4; ++q)
    memcpy(&r[q], win + lane * 16 + q * 4, 4);
  // reference (FMA) path
  float t[4][4], a[16];
  for (int q = 0; q < 4; ++q)
    cubic_1d((float)(r[q] & 255), (float)((r[q] >> 8) & 255), (float)((r[q] >> 16) & 255), (float)(r[q] >> 24), t[q][0],
             t[q][1], t[q][2], t[q][3]);
  for (int ik = 0; ik < 4; ++ik)
    cubic_1d(t[0][ik], t[1][ik], t[2][ik], t[3][ik], a[ik], a[4 + ik], a[8 + ik], a[12 + ik]);
  for (int c = 0; c < 16; ++c)
    ref[lane * 16 + c] = a[c];
  // MFMA path
  const int h = lane >> 5;
  union {
    uint32_t u[4];
    half8 v;
  } U, X;
  // rows I use (2h, 2h+1) and rows I export (2(1-h), 2(1-h)+1)
  const uint32_t u0 = h ? r[2] : r[0], u1 = h ? r[3] : r[1], x0 = h ? r[0] : r[2], x1 = h ? r[1] : r[3];
  U.u[0] = bytes_to_h2(u0, false);
  U.u[1] = bytes_to_h2(u0, true);
  U.u[2] = bytes_to_h2(u1, false);
  U.u[3] = bytes_to_h2(u1, true);
  X.u[0] = bytes_to_h2(x0, false);
  X.u[1] = bytes_to_h2(x0, true);
  X.u[2] = bytes_to_h2(x1, false);
  X.u[3] = bytes_to_h2(x1, true);
  union {
    uint32_t u[4];
    half8 v;
  } B0, B1;
  for (int q = 0; q < 4; ++q) {
    const uint32_t y = (uint32_t)__shfl_xor((int)X.u[q], 32, 64); // partner's export rows
    B0.u[q] = h == 0 ? U.u[q] : y;
    B1.u[q] = h == 0 ? y : U.u[q];
  }
  const half8 A0 = make_A(lane, 0), A1 = make_A(lane, 1);
  float16v d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0.v, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B1.v, d, 0, 0, 0);
  for (int c = 0; c < 16; ++c)
    got[lane * 16 + c] = d[c] - (c == 0 ? 1024.f : 0.f); // inputs carried the +1024 of the f16 trick
}

int main() {
  std::vector<uint8_t> win(64 * 16);
  srand(5);
  for (auto &b : win)
    b = (uint8_t)(rand() & 255);
  for (int i = 0; i < 16; ++i) { win[i] = 255; win[16 + i] = 0; } // extremes
  uint8_t *dw; float *dr, *dg;
  hipMalloc(&dw, win.size()); hipMalloc(&dr, 64 * 16 * 4); hipMalloc(&dg, 64 * 16 * 4);
  hipMemcpy(dw, win.data(), win.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dw, dr, dg);
  std::vector<float> ref(64 * 16), got(64 * 16);
  hipMemcpy(ref.data(), dr, ref.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(got.data(), dg, got.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64 * 16; ++i)
    if (memcmp(&ref[i], &got[i], 4) != 0 && bad++ < 10)
      printf("lane %d c %d ref %g got %g\n", i / 16, i % 16, ref[i], got[i]);
  printf("mismatches: %d of %d\n", bad, 64 * 16);
  return bad != 0;
}
