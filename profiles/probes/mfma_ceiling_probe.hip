// Probe (MI355X): what a bare register-operand MFMA loop sustains on RANDOM data -- the practical ceiling of an MFMA-bound
// kernel once the chip lowers its clock under load (MI355X_MICROARCH.md "DVFS give-back").  No LDS, no global traffic in
// the loop; 256 threads per block = one wave per SIMD (w1) or 512 = two waves per SIMD (w2), 16 independent accumulators
// per wave, every CU busy; run for >= 1 s per variant so the clock settles.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_ceiling_probe.hip -o mfma_ceiling_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int ET>
__global__ void __launch_bounds__(512) mfma_loop(const short8_t* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  short8_t a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  f32x4_t acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ET == 0)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a[i]), __builtin_bit_cast(half8_t, b[j]), acc[i][j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) s += acc[i][j][k];
  out[t] = s;
}

template <int ET>
double run(int threads, const short8_t* din, float* dout, int iters, int reps) {
  const int blocks = 256 * (threads == 512 ? 1 : 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<ET>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<ET>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * 16.0 * iters * reps * (2.0 * 16 * 16 * 32);
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) {                      // random finite bf16 / f16 bit patterns of moderate magnitude
    unsigned short m = rand() & 0x03ff, e = 0x3c00 + ((rand() % 5) << 10) - 0x0800, sgn = (rand() & 1) << 15;
    v = sgn | (e & 0x7c00) | m;
  }
  short8_t* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 512 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 200000;                // ~0.1 s per launch
  for (int pass = 0; pass < 2; ++pass) {
    printf("bf16 16x16x32, 1 wave/SIMD: %.0f TFLOP/s\n", run<0>(256, din, dout, iters, 12));
    printf("bf16 16x16x32, 2 waves/SIMD: %.0f TFLOP/s\n", run<0>(512, din, dout, iters / 2, 12));
    printf("f16  16x16x32, 1 wave/SIMD: %.0f TFLOP/s\n", run<1>(256, din, dout, iters, 12));
    printf("f16  16x16x32, 2 waves/SIMD: %.0f TFLOP/s\n", run<1>(512, din, dout, iters / 2, 12));
  }
  return 0;
}
