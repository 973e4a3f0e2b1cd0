// Device functions of the ant path: the reference's in-repo collision glue (bit-exact against the CPU oracle, which is pinned
// by goldens from the reference's own functions) and THIS BUILD'S stand-in dynamics model (not MuJoCo; include/ditree.h
// ditree_ant_model).  Units that include this header are compiled with -ffp-contract=off: numpy rounds every operation
// separately and the oracle restates the model operation by operation.
#pragma once
#include <hip/hip_runtime.h>

#include "fp64_device.h"

#define ANT_S 29
#define ANT_D 8

// ------------------------------------------------------------------------- collision
// common/map_utils.py:139-219 is_colliding_maze(state, maze_grid, maze_size_scaling = s, ball_radius = r) for one ball.  The
// reference returns at its first hit; the tests have no side effects, so their OR is its value.  Differences from the car's
// is_colliding_parallel (car_device.h): the cell the ball sits in is not looked at, a side test beyond the map counts as a wall
// (:177,184,191,198), a corner cell beyond the map is skipped (:212), the corner distance is math.sqrt(dx**2 + dy**2).
__device__ __forceinline__ bool ant_ball_collides(double x, double y, const unsigned char* mz, int H, int W, double s, double r) {
  const double xc = (double)W / 2.0 * s, yc = (double)H / 2.0 * s;                 // :155-156
  const double fr = floor((yc - y) / s), fc = floor((x + xc) / s);                // :158-159
  // NaN / inf -> astype(int) gives INT64_MIN -> not (0 <= row < H) -> True (:171-172)
  const bool oob = !(fr >= 0.0) | !(fr < (double)H) | !(fc >= 0.0) | !(fc < (double)W);
  const int row = oob ? 0 : (int)fr, col = oob ? 0 : (int)fc;      // (no early return: the lanes of a wave stay on one path)
  const double cell_x = ((double)col + 0.5) * s - xc, cell_y = yc - ((double)row + 0.5) * s;   // :161-162
  const double half = s / 2.0;
  const double x_min = cell_x - half, x_max = cell_x + half, y_min = cell_y - half, y_max = cell_y + half;
  // side cells read up front with clipped indices (a side beyond the map counts as a wall by the index test alone) and the
  // tests combined without short-circuit branches: the LDS reads are in flight together behind one wait
  const unsigned char m_r = mz[row * W + min(col + 1, W - 1)], m_l = mz[row * W + max(col - 1, 0)],
                      m_t = mz[max(row - 1, 0) * W + col], m_b = mz[min(row + 1, H - 1) * W + col];
  bool coll = false;
  coll |= (x + r > x_max) & ((col + 1 >= W) | (m_r == 1));                         // right  :175-178
  coll |= (x - r < x_min) & ((col - 1 < 0) | (m_l == 1));                          // left   :181-185
  coll |= (y + r > y_max) & ((row - 1 < 0) | (m_t == 1));                          // top    :188-192
  coll |= (y - r < y_min) & ((row + 1 >= H) | (m_b == 1));                         // bottom :195-199
  if (r < 0.9 * half) {
    // :202-216 tests four corners; only the distance to the NEAREST one can be below r when r is well under half a cell (the ball
    // sits in its cell up to rounding, every other corner is half a cell away along at least one axis): one sqrt, same value
    const bool right = x >= cell_x, up = y >= cell_y;
    const int ci = up ? row - 1 : row + 1, cj = right ? col + 1 : col - 1;
    const double dx = (right ? x_max : x_min) - x, dy = (up ? y_max : y_min) - y;
    const double dist = sqrt(dx * dx + dy * dy);                                    // :210
    const bool inside = (ci >= 0) & (ci < H) & (cj >= 0) & (cj < W);                 // :213
    const int i2 = min(max(ci, 0), H - 1), j2 = min(max(cj, 0), W - 1);
    coll |= (dist < r) & inside & (mz[i2 * W + j2] == 1);
    return coll | oob;
  }
  const int ci[4] = {row - 1, row - 1, row + 1, row + 1};                           // :202-207
  const int cj[4] = {col + 1, col - 1, col + 1, col - 1};
  const double kx[4] = {x_max, x_min, x_max, x_min};
  const double ky[4] = {y_max, y_max, y_min, y_min};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double dx = kx[k] - x, dy = ky[k] - y;
    const double dist = sqrt(dx * dx + dy * dy);                                    // :210
    const bool inside = ci[k] >= 0 && ci[k] < H && cj[k] >= 0 && cj[k] < W;          // :213
    const int i2 = min(max(ci[k], 0), H - 1), j2 = min(max(cj[k], 0), W - 1);
    coll |= (dist < r) && inside && (mz[i2 * W + j2] == 1);
  }
  return coll | oob;
}

// common/map_utils.py:126-136 is_colliding_ant(state, maze, ant_radius, map_scale): upside down when the body z axis points
// below the horizon, R[2][2] = 1 - 2 (qx^2 + qy^2) with (qw, qx, qy, qz) = state[3:7] (common/se3_utils.py:155-164).
__device__ __forceinline__ bool ant_collides(const double* st, const unsigned char* mz, int H, int W, double s, double r) {
  const double qx = st[4], qy = st[5];
  const double up = 1.0 - 2.0 * (qx * qx + qy * qy);
  if (up < 0.0) return true;
  return ant_ball_collides(st[0], st[1], mz, H, W, s, r);
}

// ------------------------------------------------------------------------- stand-in dynamics (NOT MuJoCo)
struct AntModelArg {
  double h;
  int frame_skip;
  double k_act, k_spr, k_dmp, k_lim, hip_lim, ank_lo, ank_hi, ank_rest, contact_gain, leg_r, k_push, c_lin, z0, z_gain, k_z,
      c_z, k_lift, c_ang, k_up, k_yaw, cphi, sphi;
};

// One env step = frame_skip sub-steps; s[29] in/out, a[8] raw (clipped to [-1, 1] here, as the env's action space does).
// Same operation order as oracle/ant.py ant_model_step.
__device__ __forceinline__ void ant_model_step(double* s, const double* a_raw, const AntModelArg& m) {
  double a[ANT_D];
#pragma unroll
  for (int k = 0; k < ANT_D; ++k) a[k] = a_raw[k] < -1.0 ? -1.0 : (a_raw[k] > 1.0 ? 1.0 : a_raw[k]);
  double x = s[0], y = s[1], z = s[2], qw = s[3], qx = s[4], qy = s[5], qz = s[6];
  double vx = s[15], vy = s[16], vz = s[17], wx = s[18], wy = s[19], wz = s[20];
  const double h = m.h;
  // mount angles 45, 135, 225, 315 degrees
  const double cph[4] = {m.cphi, -m.cphi, -m.cphi, m.cphi};
  const double sph[4] = {m.sphi, m.sphi, -m.sphi, -m.sphi};
#pragma unroll 1
  for (int it = 0; it < m.frame_skip; ++it) {
    double fxb = 0.0, fyb = 0.0, tz = 0.0, tx = 0.0, ty = 0.0, lift_sum = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      double hip = s[7 + 2 * l], ank = s[8 + 2 * l], hd = s[21 + 2 * l], ad = s[22 + 2 * l];
      double over = fmax(hip - m.hip_lim, 0.0) - fmax(-m.hip_lim - hip, 0.0);
      const double hdd = m.k_act * a[2 * l] - m.k_spr * hip - m.k_dmp * hd - m.k_lim * over;
      over = fmax(ank - m.ank_hi, 0.0) - fmax(m.ank_lo - ank, 0.0);
      const double add = m.k_act * a[2 * l + 1] - m.k_spr * (ank - m.ank_rest) - m.k_dmp * ad - m.k_lim * over;
      hd = hd + h * hdd;
      ad = ad + h * add;
      hip = hip + h * hd;
      ank = ank + h * ad;
      s[7 + 2 * l] = hip; s[8 + 2 * l] = ank; s[21 + 2 * l] = hd; s[22 + 2 * l] = ad;
      const double c = 0.5 * (1.0 + tanh_em(m.contact_gain * (ank - m.ank_rest)));
      const double push = -(m.leg_r * hd) * c;
      fxb = fxb + (-sph[l]) * push;
      fyb = fyb + cph[l] * push;
      tz = tz + m.leg_r * push;
      const double lift = c * (ank - m.ank_rest);
      lift_sum = lift_sum + lift;
      tx = tx + sph[l] * (m.k_lift * lift);
      ty = ty - cph[l] * (m.k_lift * lift);
    }
    const double yaw = atan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz));
    double cy, sy;
    sincos(yaw, &sy, &cy);
    const double axw = m.k_push * (cy * fxb - sy * fyb) - m.c_lin * vx;
    const double ayw = m.k_push * (sy * fxb + cy * fyb) - m.c_lin * vy;
    vx = vx + h * axw;
    vy = vy + h * ayw;
    x = x + h * vx;
    y = y + h * vy;
    const double z_ref = m.z0 + m.z_gain * (0.25 * lift_sum);
    vz = vz + h * (m.k_z * (z_ref - z) - m.c_z * vz);
    z = z + h * vz;
    const double u0 = 2.0 * (qx * qz - qw * qy);
    const double u1 = 2.0 * (qy * qz + qw * qx);
    wx = wx + h * (tx - m.c_ang * wx + m.k_up * (-u1));
    wy = wy + h * (ty - m.c_ang * wy + m.k_up * u0);
    wz = wz + h * (m.k_yaw * tz - m.c_ang * wz);
    const double hw = 0.5 * h;
    const double nqw = qw + hw * (-(qx * wx) - qy * wy - qz * wz);
    const double nqx = qx + hw * (qw * wx + qy * wz - qz * wy);
    const double nqy = qy + hw * (qw * wy + qz * wx - qx * wz);
    const double nqz = qz + hw * (qw * wz + qx * wy - qy * wx);
    const double inv = 1.0 / sqrt(nqw * nqw + nqx * nqx + nqy * nqy + nqz * nqz);
    qw = nqw * inv; qx = nqx * inv; qy = nqy * inv; qz = nqz * inv;
  }
  s[0] = x; s[1] = y; s[2] = z; s[3] = qw; s[4] = qx; s[5] = qy; s[6] = qz;
  s[15] = vx; s[16] = vy; s[17] = vz; s[18] = wx; s[19] = wy; s[20] = wz;
}
