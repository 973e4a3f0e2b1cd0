// FP64 device helpers shared by the dynamics headers (car_device.h, ant_device.h).
#pragma once
#include <hip/hip_runtime.h>

// tanh(x) = em / (em + 2) with em = expm1(2 |x|) >= 0 (no cancellation), |x| clamped at 20 (tanh rounds to 1 from 19.1 on), sign
// restored; NaN propagates.  At most 3 ulp from numpy's tanh over 5 M arguments in [1e-300, 25] (profiles/NOTES.md), at 41 FP64
// instructions where the device library's tanh takes 139 -- and tanh is a fifth of the car step's chain and two thirds of the
// stand-in ant step's (20 evaluations per env step).  The reference evaluates C tanh (car_env.py:380 through casadi); the library
// tanh this replaces already differed from it in the last place.
__device__ __forceinline__ double tanh_em(double x) {
  double ax = fabs(x);
  ax = ax > 20.0 ? 20.0 : ax;
  const double em = expm1(2.0 * ax);
  return copysign(em / (em + 2.0), x);
}
