// Probe (MI355X): does v_mfma_f32_16x16x32_f16 keep f16 subnormal inputs, and are its products / sums exact enough
// for a hi + lo split?  Build: hipcc --offload-arch=gfx950 -O2 f16_mfma_probe.hip -o f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// D = A(16x32) * B(32x16): lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15]
__global__ void probe(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x;
  half8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[((l >> 4) * 4 + i) * 16 + (l & 15)] = c[i];
}

int main() {
  std::vector<_Float16> A(16 * 32), B(32 * 16);
  // row 0: a single subnormal a = 2^-20 (f16 subnormal) times b = 1024 -> 2^-10 if kept, 0 if flushed
  for (auto& v : A) v = (_Float16)0.f;
  for (auto& v : B) v = (_Float16)0.f;
  A[0 * 32 + 0] = (_Float16)9.5367431640625e-07f;  // 2^-20
  B[0 * 16 + 0] = (_Float16)1024.f;
  // row 1: smallest subnormal 2^-24 times 2^10
  A[1 * 32 + 1] = (_Float16)5.9604644775390625e-08f;
  B[1 * 16 + 1] = (_Float16)1024.f;
  // row 2: subnormal on the B side
  A[2 * 32 + 2] = (_Float16)1024.f;
  B[2 * 16 + 2] = (_Float16)9.5367431640625e-07f;
  // row 3: exactness of an 11-bit x 11-bit product and of a 32-term sum with wide dynamic range
  for (int k = 0; k < 32; ++k) { A[3 * 32 + k] = (_Float16)(1.0f + k / 1024.0f); B[k * 16 + 3] = (_Float16)(1.0f - k / 2048.0f); }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  printf("subnormal A 2^-20 * 1024 = %.10g (expect %.10g)\n", D[0 * 16 + 0], 9.5367431640625e-07 * 1024);
  printf("subnormal A 2^-24 * 1024 = %.10g (expect %.10g)\n", D[1 * 16 + 1], 5.9604644775390625e-08 * 1024);
  printf("subnormal B 1024 * 2^-20 = %.10g (expect %.10g)\n", D[2 * 16 + 2], 9.5367431640625e-07 * 1024);
  double ref = 0; float reff = 0.f;
  for (int k = 0; k < 32; ++k) { double p = (double)(float)A[3 * 32 + k] * (double)(float)B[k * 16 + 3]; ref += p; reff = fmaf((float)A[3 * 32 + k], (float)B[k * 16 + 3], reff); }
  printf("32-term dot: mfma %.9g  f64 %.12g  f32-chain %.9g  rel err vs f64 %.3g\n", D[3 * 16 + 3], ref, reff, fabs(D[3 * 16 + 3] - ref) / ref);
  return 0;
}
