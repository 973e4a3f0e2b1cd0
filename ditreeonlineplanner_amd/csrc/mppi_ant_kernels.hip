// MPPI controller on the higher-DoF rollout slot (gfx950, FP64): BASELINE config 5 as written -- "MPPI antmaze, 65 536 rollouts".
//
// The reference repository ships NEITHER an MPPI module (run_scenarios_with_lidar_MPPI.py:10 imports one that is absent) NOR the
// ant's physics (MuJoCo through gymnasium-robotics).  So both the controller (this build's information-theoretic MPPI, the one
// of mppi_kernels.hip) and the dynamics (the build's stand-in crawler model, ant_device.h) are the build's own; what IS the
// reference's are the collision test (common/map_utils.py:126-219, 1.2-radius ball, upside-down test) and the goal radius
// (0.45 * s_global, planners/base_planner.py:296-297).  PARITY UNPINNED BY CONSTRUCTION, restated in oracle/mppi.py.
//
// Same structure as mppi_kernels.hip with one lane per rollout (the 29-d state and the 8-d noise leave no registers to share a
// rollout between lanes): rollouts + running minimum of the costs, fixed-order partial sums of w and w eps[k, t, 0..8), ordered
// finish + control update + one executed env step.  Noise is a counter hash (four Box-Muller pairs per (rollout, step)) that the
// update kernel regenerates: a step writes 12 bytes per rollout.
#include <algorithm>

#include "ant_device.h"
#include "ditree_internal.h"

namespace {

constexpr int AM_MAX_T = 64;
constexpr int AM_MAX_P = 4096;
constexpr int AM_SLICES = 256;
constexpr int AM_NU = 8;

__device__ __forceinline__ unsigned long long am_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct AntMppiArgs {
  int T, K, P;
  double lambda, sigma[AM_NU], w_track, w_progress, w_collision, w_goal;
  unsigned long long seed, counter;
  int wback, wfwd;
  double gx, gy, goal_radius, ball_radius, s_global;
  long long k0;
};

// eps[k, t, 0..8) ~ N(0, diag(sigma^2)): pair p = 0..3 of (seed, counter, GLOBAL k, t) gives dims 2p, 2p + 1
__device__ __forceinline__ void am_noise(const AntMppiArgs& a, long long k, int t, double* e) {
  unsigned long long h = am_splitmix64(a.seed ^ am_splitmix64(a.counter));
  h = am_splitmix64(h ^ ((unsigned long long)k * 0xD1B54A32D192ED03ull));
  h = am_splitmix64(h ^ (unsigned long long)t);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned long long h1 = am_splitmix64(h ^ ((unsigned long long)(p + 1) * 0xA24BAED4963EE407ull));
    const unsigned long long h2 = am_splitmix64(h1);
    const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586 * u2;
    double sa, ca;
    sincos(ang, &sa, &ca);
    e[2 * p] = a.sigma[2 * p] * (r * ca);
    e[2 * p + 1] = a.sigma[2 * p + 1] * (r * sa);
  }
}

__device__ __forceinline__ unsigned long long am_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double am_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ void __launch_bounds__(256)
mppi_ant_rollout_kernel(const unsigned char* __restrict__ maze, int rows, int cols, AntModelArg m, const double* __restrict__ state,
                        const double* __restrict__ U, const double2* __restrict__ path, const double* __restrict__ noise,
                        AntMppiArgs a, double* __restrict__ costs, int32_t* __restrict__ flags, double* __restrict__ result,
                        unsigned long long* __restrict__ minkey) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double2* s_path = (double2*)lds_raw;
  double* s_U = (double*)(s_path + a.P);                                 // T x 8
  double* s_red = s_U + AM_NU * a.T;
  unsigned char* s_maze = (unsigned char*)(s_red + 16);
  const int tid = threadIdx.x;
  for (int i = tid; i < a.P; i += blockDim.x) s_path[i] = path[i];
  for (int i = tid; i < AM_NU * a.T; i += blockDim.x) s_U[i] = U[i];
  for (int i = tid; i < rows * cols; i += blockDim.x) s_maze[i] = maze[i];
  const double x00 = state[0], x01 = state[1];
  __syncthreads();
  int i0;
  {
    double best = __builtin_huge_val();
    int bi = 0x7fffffff;
    for (int i = tid; i < a.P; i += blockDim.x) {
      const double dx = s_path[i].x - x00, dy = s_path[i].y - x01;
      const double d = dx * dx + dy * dy;
      if (d < best) { best = d; bi = i; }
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) {
      const double ob = __shfl_xor(best, q);
      const int oi = __shfl_xor(bi, q);
      if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    int* s_ri = (int*)(s_red + 8);
    if ((tid & 63) == 0) { s_red[tid >> 6] = best; s_ri[tid >> 6] = bi; }
    __syncthreads();
    best = s_red[0]; bi = s_ri[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (s_red[w] < best || (s_red[w] == best && s_ri[w] < bi)) { best = s_red[w]; bi = s_ri[w]; }
    i0 = bi;
  }
  const int k = blockIdx.x * blockDim.x + tid;
  if (blockIdx.x == 0 && tid == 0 && result != nullptr) result[5] = (double)i0;
  const bool live = k < a.K;
  double s[ANT_S];
#pragma unroll
  for (int j = 0; j < ANT_S; ++j) s[j] = state[j];
  double cost = 0.0;
  int ip = i0, flag = 0;
  for (int t = 0; live && t < a.T; ++t) {
    double e[AM_NU], u[AM_NU];
#pragma unroll
    for (int d = 0; d < AM_NU; ++d) e[d] = 0.0;
    if (a.k0 + k > 0) {                                // GLOBAL rollout 0 is the noise-free nominal sequence
      if (noise != nullptr) {
#pragma unroll
        for (int d = 0; d < AM_NU; ++d) e[d] = noise[((size_t)k * a.T + t) * AM_NU + d];
      } else {
        am_noise(a, a.k0 + k, t, e);
      }
    }
    double ctrl = 0.0;
#pragma unroll
    for (int d = 0; d < AM_NU; ++d) {
      const double ud = s_U[AM_NU * t + d];
      u[d] = ud + e[d];
      ctrl = ctrl + (ud * e[d]) / (a.sigma[d] * a.sigma[d]);
    }
    ant_model_step(s, u, m);                            // the model clips the action to [-1, 1] as the env's action space does
    const bool coll = ant_collides(s, s_maze, rows, cols, a.s_global, a.ball_radius);
    const double ex = s[0] - a.gx, ey = s[1] - a.gy;
    const bool reached = sqrt(fma(ey, ey, ex * ex)) < a.goal_radius;
    const int lo = max(ip - a.wback, 0), hi = min(ip + a.wfwd, a.P - 1);
    double best = __builtin_huge_val();
    int bi = lo;
    // eight points per trip: independent LDS reads and distances, only the running minimum chains (one read-wait-compare
    // round per point exposed the LDS latency 57 times per step); indices past `hi` are clamped for the read and masked
    for (int i = lo; i <= hi; i += 8) {
      double d[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double2 pt = s_path[min(i + q, hi)];
        const double dx = pt.x - s[0], dy = pt.y - s[1];
        d[q] = dx * dx + dy * dy;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i + q <= hi && d[q] < best) { best = d[q]; bi = i + q; }
    }
    ip = bi;
    cost = cost + a.w_track * best;
    cost = cost + a.lambda * ctrl;
    if (coll) { cost = cost + a.w_collision; flag = 2; break; }
    if (reached) { cost = cost - a.w_goal; flag = 1; break; }
  }
  cost = cost + a.w_progress * (double)(a.P - 1 - ip);
  if (live) {
    costs[k] = cost;
    if (flags != nullptr) flags[k] = flag;
  }
  double mn = live ? cost : __builtin_huge_val();
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o));
  if ((tid & 63) == 0) atomicMin(minkey, am_key(mn));
}

__global__ void mppi_ant_min_kernel(const unsigned long long* __restrict__ minkey, double* __restrict__ result) {
  result[3] = am_unkey(*minkey);
}

// Grid (slices, T): block (sl, t) reduces step t of slice sl's rollouts (block t = 0 also the three scalar sums and the weights).
// One block per slice walked all T steps before: 1 024 waves on 1 024 SIMDs, each regenerating 64 Box-Muller pairs per rollout
// with nothing to hide their latency (78 us at K = 65 536, T = 16).  Same sums in the same order.
__global__ void __launch_bounds__(256)
mppi_ant_partial_kernel(const double* __restrict__ costs, const double* __restrict__ noise, AntMppiArgs a, const double* __restrict__ result,
                        double* __restrict__ partial /*[slices][3 + 8T]*/, double* __restrict__ weights, const int32_t* __restrict__ flags) {
  __shared__ double red[4][3 + AM_NU];
  const int t = blockIdx.y;
  const int per = (a.K + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(lo + per, a.K);
  const double beta = result[3];
  const int nacc = 3 + AM_NU * a.T;
  const int wv = threadIdx.x >> 6;
  const bool lane0 = (threadIdx.x & 63) == 0;
  if (threadIdx.x < 4 * (3 + AM_NU)) (&red[0][0])[threadIdx.x] = 0.0;
  __syncthreads();
  auto wave_sum = [](double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  for (int kb = lo; kb < hi; kb += 256) {
    const int k = kb + (int)threadIdx.x;
    const bool valid = k < hi;
    const double w = valid ? exp(-(costs[k] - beta) / a.lambda) : 0.0;
    if (t == 0) {
      if (valid && weights != nullptr) weights[k] = w;          // un-normalised; mppi_ant_weights_kernel divides by eta
      const double sw = wave_sum(w), sw2 = wave_sum(w * w);
      const double sc = wave_sum((valid && flags != nullptr && flags[k] == 2) ? 1.0 : 0.0);
      if (lane0) { red[wv][0] += sw; red[wv][1] += sw2; red[wv][2] += sc; }
    }
    double e[AM_NU];
#pragma unroll
    for (int d = 0; d < AM_NU; ++d) e[d] = 0.0;
    if (valid && a.k0 + k > 0) {
      if (noise != nullptr) {
#pragma unroll
        for (int d = 0; d < AM_NU; ++d) e[d] = noise[((size_t)k * a.T + t) * AM_NU + d];
      } else {
        am_noise(a, a.k0 + k, t, e);
      }
    }
#pragma unroll
    for (int d = 0; d < AM_NU; ++d) {
      const double v = wave_sum(w * e[d]);
      if (lane0) red[wv][3 + d] += v;
    }
  }
  __syncthreads();
  if (t == 0 && threadIdx.x < 3)
    partial[(size_t)blockIdx.x * nacc + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x < AM_NU) {
    const int j = 3 + threadIdx.x;
    partial[(size_t)blockIdx.x * nacc + 3 + AM_NU * t + threadIdx.x] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
}

__global__ void __launch_bounds__(256) mppi_ant_weights_kernel(double* __restrict__ weights, const double* __restrict__ sums, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) weights[k] = weights[k] / sums[0];
}

// result [64]: executed action 0..7 at [16..24), status [2], beta [3], eta [4], nearest path index [5], collided [6], effective
// samples [7], the state after EXECUTE at [24..53)
__global__ void __launch_bounds__(256)
mppi_ant_finish_kernel(const unsigned char* __restrict__ maze, int rows, int cols, AntModelArg m, const double* __restrict__ partial,
                       int slices, AntMppiArgs a, double* __restrict__ state_io, double* __restrict__ U, double* __restrict__ weights,
                       double* __restrict__ sums, double* __restrict__ result, int do_sums, int do_apply, int do_execute) {
  const int nacc = 3 + AM_NU * a.T;
  if (do_sums) {
    // thread j adds value j of the slices in slice order (one fixed order: reproducible); 3 + 8 T values, <= 256 slices --
    // the loads of a value are independent, only the adds chain (a per-value block reduction with two barriers each took
    // 98 us for the 131 values of T = 16: profiles/r04_mppi_ant_bench_kernel_stats.csv)
    for (int j = threadIdx.x; j < nacc; j += 256) {
      double acc = 0.0;
      for (int s0 = 0; s0 < slices; s0 += 16) {                 // 16 loads in flight, then their adds in slice order
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s0 + u < slices ? partial[(size_t)(s0 + u) * nacc + j] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
      sums[j] = acc;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (do_apply) {
    const double eta = sums[0];
    for (int j = threadIdx.x; j < AM_NU * a.T; j += 256) U[j] = U[j] + sums[3 + j] / eta;
    if (threadIdx.x == 0) {                                   // (the weights are normalised by mppi_ant_weights_kernel)
      result[4] = eta;
      result[6] = sums[2];
      result[7] = (eta * eta) / sums[1];
    }
    __syncthreads();
  }
  if (do_execute && threadIdx.x == 0) {
    double s[ANT_S], u[AM_NU];
    for (int j = 0; j < ANT_S; ++j) s[j] = state_io[j];
    for (int d = 0; d < AM_NU; ++d) u[d] = fmin(fmax(U[d], -1.0), 1.0);
    ant_model_step(s, u, m);
    const double ex = s[0] - a.gx, ey = s[1] - a.gy;
    const bool reached = sqrt(fma(ey, ey, ex * ex)) < a.goal_radius;
    const bool coll = ant_collides(s, maze, rows, cols, a.s_global, a.ball_radius);
    for (int d = 0; d < AM_NU; ++d) result[16 + d] = u[d];
    if (coll) {
      result[2] = 2.0;
      for (int j = 0; j < AM_NU * a.T; ++j) U[j] = 0.0;
      for (int j = 0; j < ANT_S; ++j) result[24 + j] = state_io[j];
    } else {
      result[2] = reached ? 1.0 : 0.0;
      for (int j = 0; j < ANT_S; ++j) { state_io[j] = s[j]; result[24 + j] = s[j]; }
      for (int t = 0; t + 1 < a.T; ++t)
        for (int d = 0; d < AM_NU; ++d) U[AM_NU * t + d] = U[AM_NU * (t + 1) + d];
    }
  }
}

}  // namespace

extern "C" int32_t ditree_mppi_step_ant(ditree_ctx* ctx, const ditree_mppi_ant_params* p, double* state_io, double* U_io,
                                        const double* path_xy, int32_t P, const double* desired_goal_xy, const double* noise,
                                        uint64_t counter, int32_t stages, double* costs, double* weights, int32_t* flags,
                                        double* sums, double* result, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!ctx->maze) return set_err(ctx, DITREE_E_STATE, "mppi_step_ant: no maze uploaded");
  if (!p || !state_io || !U_io || !path_xy || !desired_goal_xy || !costs || !result || !sums || p->T < 1 || p->T > AM_MAX_T ||
      p->K < 1 || P < 1 || P > AM_MAX_P || !(p->lambda > 0.0) || p->window_back < 0 || p->window_fwd < 0 ||
      (stages & ~DITREE_MPPI_ALL) != 0 || stages == 0 || p->k_offset < 0 || !(p->s_global > 0.0))
    return set_err(ctx, DITREE_E_ARG, "mppi_step_ant: bad argument (1 <= T <= 64, 1 <= P <= 4096, lambda > 0, stages 1..31)");
  for (int d = 0; d < AM_NU; ++d)
    if (!(p->sigma[d] > 0.0)) return set_err(ctx, DITREE_E_ARG, "mppi_step_ant: sigma must be positive");
  if (!(p->model.h > 0.0) || !(p->model.frame_skip >= 1.0) || p->model.frame_skip > 64.0 || p->model.frame_skip != (double)(int)p->model.frame_skip)
    return set_err(ctx, DITREE_E_ARG, "mppi_step_ant: model needs h > 0 and an integral frame_skip in 1..64");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const ditree_ant_model& mm = p->model;
  AntModelArg m;
  m.h = mm.h; m.frame_skip = (int)mm.frame_skip; m.k_act = mm.k_act; m.k_spr = mm.k_spr; m.k_dmp = mm.k_dmp; m.k_lim = mm.k_lim;
  m.hip_lim = mm.hip_lim; m.ank_lo = mm.ank_lo; m.ank_hi = mm.ank_hi; m.ank_rest = mm.ank_rest; m.contact_gain = mm.contact_gain;
  m.leg_r = mm.leg_r; m.k_push = mm.k_push; m.c_lin = mm.c_lin; m.z0 = mm.z0; m.z_gain = mm.z_gain; m.k_z = mm.k_z; m.c_z = mm.c_z;
  m.k_lift = mm.k_lift; m.c_ang = mm.c_ang; m.k_up = mm.k_up; m.k_yaw = mm.k_yaw; m.cphi = mm.cphi; m.sphi = mm.sphi;
  AntMppiArgs a;
  a.T = p->T; a.K = p->K; a.P = P;
  a.lambda = p->lambda;
  for (int d = 0; d < AM_NU; ++d) a.sigma[d] = p->sigma[d];
  a.w_track = p->w_track; a.w_progress = p->w_progress; a.w_collision = p->w_collision; a.w_goal = p->w_goal;
  a.seed = p->seed; a.counter = counter;
  a.wback = p->window_back; a.wfwd = p->window_fwd;
  a.gx = desired_goal_xy[0]; a.gy = desired_goal_xy[1];
  a.goal_radius = p->goal_radius; a.ball_radius = p->ball_radius; a.s_global = p->s_global;
  a.k0 = p->k_offset;
  const int slices = std::min(AM_SLICES, (a.K + 255) / 256);
  if (!ctx->mppi_ant_partial || !ctx->mppi_minkey) {
    if (!ctx->mppi_ant_partial)
      HIP_TRY(ctx, hipMalloc((void**)&ctx->mppi_ant_partial, (size_t)AM_SLICES * (3 + AM_NU * AM_MAX_T) * sizeof(double)));
    if (!ctx->mppi_minkey) HIP_TRY(ctx, hipMalloc((void**)&ctx->mppi_minkey, sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->mppi_minkey, 0xFF, sizeof(unsigned long long), s));
  }
  if (stages & DITREE_MPPI_ROLLOUTS) {
    const size_t lds = (size_t)P * 16 + (size_t)a.T * AM_NU * 8 + 128 + (((size_t)ctx->rows * ctx->cols + 15) & ~(size_t)15);
    if (lds > 160 * 1024) return set_err(ctx, DITREE_E_ARG, "mppi_step_ant: path + maze exceed the LDS");
    static bool attr_done[64] = {};
    if (lds > 64 * 1024 && ctx->device < 64 && !attr_done[ctx->device]) {
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)mppi_ant_rollout_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[ctx->device] = true;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->mppi_minkey, 0xFF, sizeof(unsigned long long), s));
    const int blk = a.K >= 32768 ? 256 : 64;
    hipLaunchKernelGGL(mppi_ant_rollout_kernel, dim3((a.K + blk - 1) / blk), dim3(blk), lds, s, ctx->maze, ctx->rows, ctx->cols, m,
                       state_io, U_io, (const double2*)path_xy, noise, a, costs, flags, result, ctx->mppi_minkey);
  }
  if (stages & DITREE_MPPI_MIN) hipLaunchKernelGGL(mppi_ant_min_kernel, dim3(1), dim3(1), 0, s, ctx->mppi_minkey, result);
  if (stages & DITREE_MPPI_SUMS)
    hipLaunchKernelGGL(mppi_ant_partial_kernel, dim3(slices, a.T), dim3(256), 0, s, costs, noise, a, result, ctx->mppi_ant_partial, weights, flags);
  if (stages & (DITREE_MPPI_SUMS | DITREE_MPPI_APPLY | DITREE_MPPI_EXECUTE))
    hipLaunchKernelGGL(mppi_ant_finish_kernel, dim3(1), dim3(256), 0, s, ctx->maze, ctx->rows, ctx->cols, m, ctx->mppi_ant_partial, slices,
                       a, state_io, U_io, weights, sums, result, (stages & DITREE_MPPI_SUMS) ? 1 : 0,
                       (stages & DITREE_MPPI_APPLY) ? 1 : 0, (stages & DITREE_MPPI_EXECUTE) ? 1 : 0);
  if ((stages & DITREE_MPPI_APPLY) && weights != nullptr)
    hipLaunchKernelGGL(mppi_ant_weights_kernel, dim3((a.K + 255) / 256), dim3(256), 0, s, weights, sums, a.K);
  HIP_TRY(ctx, hipGetLastError());
  return DITREE_OK;
}
