// Device functions of the car model shared by the translation units that integrate it (geom_kernels.hip: rollouts of the
// expansion rounds and the online plan follower; mppi_kernels.hip: the MPPI controller's rollouts).  Every unit that includes
// this header is compiled with -ffp-contract=off: the reference evaluates these expressions with separately rounded numpy
// operations and collision / goal flags must be bit-exact against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "fp64_device.h"

// ------------------------------------------------------------------------- collision
// common/map_utils.py:221-329 for one ball (maze_size_scaling = 1, radius 0.1).
__device__ __forceinline__ bool ball_collides(double x, double y, const unsigned char* mz, int H, int W) {
  const double s = 1.0, r = 0.1;
  const double xc = (double)W / 2.0 * s, yc = (double)H / 2.0 * s;
  double fr = floor((yc - y) / s), fc = floor((x + xc) / s);
  // numpy's astype(int) maps NaN/inf to INT64_MIN -> out of bounds -> collision (:255-259)
  const bool oob = !(fr >= 0.0) | !(fr < (double)H) | !(fc >= 0.0) | !(fc < (double)W);
  const int row = oob ? 0 : (int)fr, col = oob ? 0 : (int)fc;      // (no early return: the lanes of a wave stay on one path)
  const double cell_x = ((double)col + 0.5) * s - xc, cell_y = yc - ((double)row + 0.5) * s;
  const double half = s / 2.0;
  const double x_min = cell_x - half, x_max = cell_x + half, y_min = cell_y - half, y_max = cell_y + half;
  const int cr = min(col + 1, W - 1), cl = max(col - 1, 0), rt = max(row - 1, 0), rb = min(row + 1, H - 1);
  // corners :315-327; invalid neighbour => collision; column clipped with map_length (sic, :326).
  // The reference tests all four corners: `invalid or (hypot(corner - p) < r and occupied)`.  Any invalid corner <=> the cell is on the
  // map's border.  Of the four distances only the one to the NEAREST corner can be below r: the ball sits in its cell (up to
  // rounding), every other corner is half a cell (0.5 >> r = 0.1) away along at least one axis and hypot >= max(|dx|, |dy|).
  // One hypot instead of four, same value.
  const bool right = x >= cell_x, up = y >= cell_y;
  const int ci = up ? row - 1 : row + 1, cj = right ? col + 1 : col - 1;
  const int i2 = min(max(ci, 0), H - 1);
  const int j2 = min(min(max(cj, 0), H - 1), W - 1);
  // the six cells are read up front (every index is clipped into the map) and the tests are combined without short-circuit
  // branches: six LDS reads in flight behind one wait instead of six dependent read-wait-branch rounds
  const unsigned char m_c = mz[row * W + col], m_r = mz[row * W + cr], m_l = mz[row * W + cl], m_t = mz[rt * W + col],
                      m_b = mz[rb * W + col], m_k = mz[i2 * W + j2];
  const double dist = hypot((right ? x_max : x_min) - x, (up ? y_max : y_min) - y);
  bool coll = m_c == 1;                                                          // :262
  coll |= (x + r > x_max) & (m_r == 1);                                          // right  :288-292
  coll |= (x - r < x_min) & (m_l == 1);                                          // left   :294-298
  coll |= (y + r > y_max) & (m_t == 1);                                          // top    :300-304
  coll |= (y - r < y_min) & (m_b == 1);                                          // bottom :306-310
  coll |= (row == 0) | (row == H - 1) | (col == 0) | (col == W - 1);
  coll |= (dist < r) & (m_k == 1);
  return coll | oob;
}

// common/map_utils.py:103-115: two balls +-0.075 m along the heading.
__device__ __forceinline__ bool car_collides(double x, double y, double psi, const unsigned char* mz, int H, int W) {
  const double off = 0.15 * 0.5;
  double sps, cps;
  sincos(psi, &sps, &cps);       // one range reduction for the pair; ocml's sin / cos evaluate the same kernels (identical values)
  double ox = off * cps, oy = off * sps;
  bool f = ball_collides(x + ox, y + oy, mz, H, W);
  bool b = ball_collides(x - ox, y - oy, mz, H, W);
  return f || b;
}

// car_env.py:356-396 _update_state: clip the action, explicit dynamics, one Euler step of 1/50 s.
__device__ __forceinline__ void car_euler_step(double* s, double a0r, double a1r) {
  // car_env.py:32,47-53
  const double dt = 1.0 / 50.0, m = 0.043, C1 = 0.5, C2 = 15.5, Cm1 = 0.28, Cm2 = 0.05, Cr0 = 0.011, Cr2 = 0.006;
  // np.clip(action, [-10,-2], [10,2]) car_env.py:371; NaN propagates like numpy
  double a0 = a0r < -10.0 ? -10.0 : (a0r > 10.0 ? 10.0 : a0r);
  double a1 = a1r < -2.0 ? -2.0 : (a1r > 2.0 ? 2.0 : a1r);
  const double psi = s[2], v = s[3], D = s[4], dl = s[5];
  double Fxd = (Cm1 - Cm2 * v) * D - Cr2 * (v * v) - Cr0 * tanh_em(5.0 * v);     // :380
  double ang = psi + C1 * dl;
  double sang, cang;
  sincos(ang, &sang, &cang);
  double d0 = v * cang, d1 = v * sang, d2 = v * C2 * dl, d3 = (Fxd / m) * cos(C1 * dl);
  s[0] = s[0] + dt * d0;
  s[1] = s[1] + dt * d1;
  s[2] = s[2] + dt * d2;
  s[3] = s[3] + dt * d3;
  s[4] = s[4] + dt * a0;
  s[5] = s[5] + dt * a1;
}

