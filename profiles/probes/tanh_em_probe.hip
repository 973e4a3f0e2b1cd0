// tanh_em (csrc/fp64_device.h) against the device library's tanh: largest difference in units of the last place over 2^26 arguments,
// |x| log-uniform in [1e-300, 30], both signs.  hipcc --offload-arch=gfx950 -O3 -I ditreeonlineplanner_amd/csrc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "fp64_device.h"
__global__ void probe(unsigned long long* out, int n_per) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long worst = 0, differ = 0;
  for (int r = 0; r < n_per; ++r) {
    uint64_t z = (i * (uint64_t)n_per + r) * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const double mag = exp(log(1e-300) + u * (log(30.0) - log(1e-300)));
    const double x = (z & 1) ? -mag : mag;
    const long long a = __double_as_longlong(fabs(tanh_em(x))), b = __double_as_longlong(fabs(tanh(x)));
    const unsigned long long d = (unsigned long long)(a > b ? a - b : b - a);
    worst = d > worst ? d : worst;
    differ += d != 0;
  }
  atomicMax(out, worst);
  atomicAdd(out + 1, differ);
}
int main() {
  unsigned long long* d; unsigned long long h[2] = {0, 0};
  (void)hipMalloc(&d, 16); (void)hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d, 64);
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("{\"arguments\": %llu, \"max_ulp_difference_from_library_tanh\": %llu, \"arguments_that_differ\": %llu}\n", 4096ull * 256 * 64, h[0], h[1]);
  return 0;
}
