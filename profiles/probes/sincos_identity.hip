// Are ocml's sin(x) / cos(x) the same doubles as sincos(x)?  (The rollout kernels take both values of a pair from one sincos call.)
// Counts bit-level mismatches over 2^26 arguments spread over |x| in [1e-8, 1e9], both signs.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(unsigned long long* bad, int n_per) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  for (int r = 0; r < n_per; ++r) {
    uint64_t z = (i * (uint64_t)n_per + r) * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const double mag = exp(log(1e-8) + u * (log(1e9) - log(1e-8)));
    const double x = (z & 1) ? -mag : mag;
    double s, c;
    sincos(x, &s, &c);
    const double s1 = sin(x), c1 = cos(x);
    mine += (__double_as_longlong(s) != __double_as_longlong(s1)) + (__double_as_longlong(c) != __double_as_longlong(c1));
  }
  if (mine) atomicAdd(bad, mine);
}
int main() {
  unsigned long long* d; unsigned long long h = 0;
  hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d, 64);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("{\"arguments\": %llu, \"sin_or_cos_differs_from_sincos\": %llu}\n", 4096ull * 256 * 64, h);
  return h != 0;
}
