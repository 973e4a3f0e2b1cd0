// MPPI controller on the car model (gfx950, FP64): K x T rollouts, costs, soft-min weights, control update, one executed
// step -- the controller `run_scenarios_with_lidar_MPPI.py:10,339-449` imports as MPPI.mppi.MPPI.  The reference repository
// does NOT contain that module: cost function, sampling scheme and update rule below are this build's own
// (information-theoretic MPPI, Williams et al. 2017, on the reference's car dynamics / collision / goal functions), stated in
// DESIGN.md and restated on the CPU in oracle/mppi.py; PARITY WITH THE REFERENCE IS UNPINNED by construction.
//
// Compiled with -ffp-contract=off like geom_kernels.hip (same car_device.h functions, bit-exact collision / goal flags).
//
//   rollouts   mppi_rollout_kernel<G>: G = 4 adjacent lanes (a DPP quad) share one rollout.  A rollout is a sequential FP64
//              chain (T x [dynamics, two-ball collision, goal test, nearest reference-path point]); 65 536 rollouts on one lane
//              each are 1 024 waves = ONE wave per SIMD, which leaves the chain's latencies exposed (the round-2 finding on
//              car_rollout_kernel).  With four lanes per rollout there are four waves per SIMD; every lane integrates the
//              (identical) dynamics, lane g tests ball g & 1 and scans every 4th point of the path window, and the quad
//              combines by DPP (no LDS, no barrier).  Noise is not read from HBM: it is a counter-based hash
//              (splitmix64 -> Box-Muller) of (seed, call counter, rollout, step) that the update kernel regenerates, so a
//              step moves K x 8 bytes (the costs) instead of K x T x 16.  A noise tape in HBM can be passed for tests.
//   update     mppi_partial_kernel: fixed-order partial sums of w_k and w_k eps[k, t, :] over slices of the rollouts;
//              mppi_finish_kernel: ordered sum of the partials, U += sum / eta, then (stage 4) one executed env step with
//              U[0] and the shift of the nominal sequence.  All sums have a fixed order: results are reproducible.
#include <algorithm>

#include "car_device.h"
#include "ditree_internal.h"

namespace {

constexpr int MPPI_MAX_T = 64;
constexpr int MPPI_MAX_P = 4096;         // reference path points staged in LDS (64 KB of f64 pairs)
constexpr int MPPI_SLICES = 256;         // partial-sum slices of the update
constexpr int MPPI_DEFAULT_LANES = 2;    // lanes per rollout when the caller does not choose (measured: DESIGN.md "MPPI")

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// eps[k, t, :] ~ N(0, diag(sigma^2)): counter-based, a pure function of (seed, counter, k, t)
__device__ __forceinline__ void mppi_noise(unsigned long long seed, unsigned long long counter, long long k, int t, double s0,
                                           double s1, double& e0, double& e1) {
  unsigned long long h = splitmix64(seed ^ splitmix64(counter));
  h = splitmix64(h ^ ((unsigned long long)k * 0xD1B54A32D192ED03ull));
  h = splitmix64(h ^ (unsigned long long)t);
  const unsigned long long h2 = splitmix64(h);
  const double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);      // (0, 1]
  const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);             // [0, 1)
  const double r = sqrt(-2.0 * log(u1));
  const double a = 6.283185307179586 * u2;
  double sa, ca;
  sincos(a, &sa, &ca);
  e0 = s0 * (r * ca);
  e1 = s1 * (r * sa);
}

// order-preserving map double -> unsigned 64-bit (and back): negative values flip all bits, the others the sign bit
__device__ __forceinline__ unsigned long long mppi_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double mppi_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

struct MppiArgs {
  int T, K, P;
  double lambda, s0, s1, w_track, w_progress, w_collision, w_goal;
  unsigned long long seed, counter;
  int wback, wfwd;
  double gx, gy;
  long long k0;                       // global index of this rank's first rollout (sharded controller)
};

template <int G> __device__ __forceinline__ double quad_min(double v) {
  if constexpr (G >= 2) v = fmin(v, __shfl_xor(v, 1));
  if constexpr (G == 4) v = fmin(v, __shfl_xor(v, 2));
  return v;
}
template <int G> __device__ __forceinline__ int quad_min_i(int v) {
  if constexpr (G >= 2) v = min(v, __shfl_xor(v, 1));
  if constexpr (G == 4) v = min(v, __shfl_xor(v, 2));
  return v;
}
template <int G> __device__ __forceinline__ bool quad_or(bool v) {
  if constexpr (G >= 2) {
    int x = v ? 1 : 0;
    x |= __shfl_xor(x, 1);
    if constexpr (G == 4) x |= __shfl_xor(x, 2);
    return x != 0;
  }
  return v;
}

// nearest path index of `x, y` in [lo, hi] (first occurrence of the minimum, as np.argmin); lane g scans lo + g, lo + g + G, ...
template <int G>
__device__ __forceinline__ void nearest_in_window(const double2* path, int lo, int hi, double x, double y, int g, int& idx,
                                                  double& d2) {
  double best = __builtin_huge_val();
  int bi = 0x7fffffff;
  // four points per trip: their LDS reads and distances are independent (one read-wait-compare round per point left the LDS
  // latency exposed 28 times per step -- the window search was the longest part of a controller step), only the running
  // minimum chains; indices past `hi` are clamped for the read and masked for the compare
  for (int i = lo + g; i <= hi; i += 4 * G) {
    double d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 p = path[min(i + u * G, hi)];
      const double dx = p.x - x, dy = p.y - y;
      d[u] = dx * dx + dy * dy;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * G <= hi && d[u] < best) { best = d[u]; bi = i + u * G; }
  }
  const double m = quad_min<G>(best);
  // first index attaining the minimum over the quad (ties -> lowest index)
  idx = quad_min_i<G>(best == m ? bi : 0x7fffffff);
  d2 = m;
}

template <int G>
__global__ void __launch_bounds__(256)
mppi_rollout_kernel(const unsigned char* __restrict__ maze, int rows, int cols, const double* __restrict__ state,
                    const double* __restrict__ U, const double2* __restrict__ path, const double* __restrict__ noise,
                    MppiArgs a, double* __restrict__ costs, int32_t* __restrict__ flags, double* __restrict__ result,
                    unsigned long long* __restrict__ minkey) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double2* s_path = (double2*)lds_raw;                                   // P points
  double* s_U = (double*)(s_path + a.P);                                 // T x 2
  double* s_red = s_U + 2 * a.T;                                         // 2 x 4 (block argmin)
  unsigned char* s_maze = (unsigned char*)(s_red + 16);
  const int tid = threadIdx.x;
  for (int i = tid; i < a.P; i += blockDim.x) s_path[i] = path[i];
  for (int i = tid; i < 2 * a.T; i += blockDim.x) s_U[i] = U[i];
  for (int i = tid; i < rows * cols; i += blockDim.x) s_maze[i] = maze[i];
  double x0[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) x0[j] = state[j];
  __syncthreads();
  // i0 = nearest path point of the current state over the WHOLE path (block-cooperative, fixed order -> same in every block)
  int i0;
  {
    double best = __builtin_huge_val();
    int bi = 0x7fffffff;
    for (int i = tid; i < a.P; i += blockDim.x) {
      const double dx = s_path[i].x - x0[0], dy = s_path[i].y - x0[1];
      const double d = dx * dx + dy * dy;
      if (d < best) { best = d; bi = i; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double ob = __shfl_xor(best, m);
      const int oi = __shfl_xor(bi, m);
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
  const int k = (blockIdx.x * blockDim.x + tid) / G, g = tid & (G - 1);
  if (blockIdx.x == 0 && tid == 0 && result != nullptr) result[5] = (double)i0;
  const bool live = k < a.K;                           // quad-uniform (G lanes share k)
  double s[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) s[j] = x0[j];
  double cost = 0.0;
  int ip = i0;
  int flag = 0;                                        // 1 = reached the goal, 2 = collided
  double n0 = 0.0, n1 = 0.0;                           // G >= 2: the noise this lane generated for step (t & ~1) + (g & 1)
  for (int t = 0; live && t < a.T; ++t) {
    double e0 = 0.0, e1 = 0.0;
    if (a.k0 + k > 0) {                                // GLOBAL rollout 0 is the noise-free nominal sequence
      if (noise != nullptr) {
        e0 = noise[((size_t)k * a.T + t) * 2]; e1 = noise[((size_t)k * a.T + t) * 2 + 1];
      } else if constexpr (G >= 2) {
        // the lanes of a rollout would all evaluate the same Box-Muller pair (log, sqrt, sincos: a quarter of the step): lane
        // parity p generates the pair of step t + p on even t and hands it to its neighbour for the other step (the branches
        // around this are uniform over the lanes of a rollout, so both partners always execute the exchange)
        if ((t & 1) == 0) {
          n0 = 0.0; n1 = 0.0;
          if (t + (g & 1) < a.T) mppi_noise(a.seed, a.counter, a.k0 + k, t + (g & 1), a.s0, a.s1, n0, n1);
        }
        const double o0 = __shfl_xor(n0, 1), o1 = __shfl_xor(n1, 1);
        const bool mine = (t & 1) == (g & 1);
        e0 = mine ? n0 : o0; e1 = mine ? n1 : o1;
      } else {
        mppi_noise(a.seed, a.counter, a.k0 + k, t, a.s0, a.s1, e0, e1);
      }
    }
    const double u0 = s_U[2 * t], u1 = s_U[2 * t + 1];
    car_euler_step(s, u0 + e0, u1 + e1);
    // collision: lane g tests ball g & 1 (front / back), the quad ORs
    bool coll;
    if constexpr (G >= 2) {
      const double off = 0.15 * 0.5;
      const double sgn = (g & 1) ? -1.0 : 1.0;
      double sps, cps;
      sincos(s[2], &sps, &cps);
      const double ox = off * cps, oy = off * sps;
      coll = quad_or<G>(ball_collides(s[0] + sgn * ox, s[1] + sgn * oy, s_maze, rows, cols));
    } else {
      coll = car_collides(s[0], s[1], s[2], s_maze, rows, cols);
    }
    const double ex = s[0] - a.gx, ey = s[1] - a.gy;
    const bool reached = sqrt(fma(ey, ey, ex * ex)) < 0.5;                       // car_env.py:346-351
    const int lo = max(ip - a.wback, 0), hi = min(ip + a.wfwd, a.P - 1);
    double d2;
    nearest_in_window<G>(s_path, lo, hi, s[0], s[1], g, ip, d2);
    // information-theoretic MPPI: stage cost q(x) + lambda u^T Sigma^-1 eps
    cost = cost + a.w_track * d2;
    cost = cost + a.lambda * ((u0 * e0) / (a.s0 * a.s0) + (u1 * e1) / (a.s1 * a.s1));
    if (coll) { cost = cost + a.w_collision; flag = 2; break; }
    if (reached) { cost = cost - a.w_goal; flag = 1; break; }
  }
  cost = cost + a.w_progress * (double)(a.P - 1 - ip);
  if (live && g == 0) {
    costs[k] = cost;
    if (flags != nullptr) flags[k] = flag;
  }
  // beta = min_k S_k without a second pass over the costs: wave minimum, one atomicMin per wave on an order-preserving
  // integer image of the double (min is order-independent: reproducible)
  double m = (live && g == 0) ? cost : __builtin_huge_val();
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmin(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) atomicMin(minkey, mppi_key(m));
}

// ---- update: beta = min S, w_k = exp(-(S_k - beta) / lambda), eta = sum w, dU[t] = sum_k w_k eps[k, t] / eta
__global__ void mppi_min_kernel(const unsigned long long* __restrict__ minkey, double* __restrict__ result) {
  result[3] = mppi_unkey(*minkey);                    // the rollout kernel's atomicMin, decoded
}

// Grid (slices, T): slice s owns rollouts [s * per, (s + 1) * per), block (s, t) reduces step t of them (block t = 0 also the
// three scalar sums and the weights).  A thread takes rollout lo + tid (+ 256, ...): the products w eps are reduced over the
// wave by a fixed butterfly and lane 0 adds them to its wave's LDS row in loop order, the four rows are added in a fixed tree:
// every sum has one order, results are reproducible.  (One block per slice walked all T steps before: one wave per SIMD with
// nothing to hide the latency of the regenerated Box-Muller noise.)
__global__ void __launch_bounds__(256)
mppi_partial_kernel(const double* __restrict__ costs, const double* __restrict__ noise, MppiArgs a, const double* __restrict__ result,
                    double* __restrict__ partial /*[slices][3 + 2T]: eta, sum w^2, collided, sum w eps*/, double* __restrict__ weights,
                    const int32_t* __restrict__ flags) {
  __shared__ double red[4][5];
  const int t = blockIdx.y;
  const int per = (a.K + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(lo + per, a.K);
  const double beta = result[3];
  const int nacc = 3 + 2 * a.T;
  const int wv = threadIdx.x >> 6;
  const bool lane0 = (threadIdx.x & 63) == 0;
  if (threadIdx.x < 20) (&red[0][0])[threadIdx.x] = 0.0;
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
      if (valid && weights != nullptr) weights[k] = w;            // un-normalised; mppi_weights_kernel divides by eta
      const double sw = wave_sum(w), sw2 = wave_sum(w * w);
      const double sc = wave_sum((valid && flags != nullptr && flags[k] == 2) ? 1.0 : 0.0);      // collided rollouts (exact: integers)
      if (lane0) { red[wv][0] += sw; red[wv][1] += sw2; red[wv][2] += sc; }
    }
    double e0 = 0.0, e1 = 0.0;
    if (valid && a.k0 + k > 0) {
      if (noise != nullptr) { e0 = noise[((size_t)k * a.T + t) * 2]; e1 = noise[((size_t)k * a.T + t) * 2 + 1]; }
      else mppi_noise(a.seed, a.counter, a.k0 + k, t, a.s0, a.s1, e0, e1);
    }
    const double v0 = wave_sum(w * e0), v1 = wave_sum(w * e1);
    if (lane0) { red[wv][3] += v0; red[wv][4] += v1; }
  }
  __syncthreads();
  if (t == 0 && threadIdx.x < 3)
    partial[(size_t)blockIdx.x * nacc + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x < 2) {
    const int j = 3 + threadIdx.x;
    partial[(size_t)blockIdx.x * nacc + 3 + 2 * t + threadIdx.x] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
}

__global__ void __launch_bounds__(256) mppi_weights_kernel(double* __restrict__ weights, const double* __restrict__ sums, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) weights[k] = weights[k] / sums[0];
}

// sums [3 + 2T] = {eta = sum w, sum w^2, collided rollouts, sum_k w_k eps[k, t, d]}: the quantities a sharded controller
// all-reduces (SUM) between its ranks.  do_sums: ordered sum of the slices -> sums;  do_apply: U += sums[3..] / eta, weights
// normalised, result[4], [6], [7];  do_execute: one env step with U[0], shift.
__global__ void __launch_bounds__(256)
mppi_finish_kernel(const unsigned char* __restrict__ maze, int rows, int cols, const double* __restrict__ partial, int slices,
                   MppiArgs a, double* __restrict__ state_io, double* __restrict__ U, double* __restrict__ weights,
                   double* __restrict__ sums, double* __restrict__ result, int do_sums, int do_apply, int do_execute) {
  const int nacc = 3 + 2 * a.T;
  if (do_sums) {
    // thread j adds value j of the slices in slice order (one fixed order: reproducible); 3 + 2 T values, <= 256 slices
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
    for (int j = threadIdx.x; j < 2 * a.T; j += 256) U[j] = U[j] + sums[3 + j] / eta;
    if (threadIdx.x == 0) {
      result[4] = eta;
      result[6] = sums[2];
      result[7] = (eta * eta) / sums[1];                                        // effective sample size
    }
    __syncthreads();
  }
  if (do_execute && threadIdx.x == 0) {
    // one env step with the first control (clipped as the env clips it, car_env.py:371), collision against the KNOWN maze
    double s[6];
    for (int j = 0; j < 6; ++j) s[j] = state_io[j];
    const double a0 = fmin(fmax(U[0], -10.0), 10.0), a1 = fmin(fmax(U[1], -2.0), 2.0);
    car_euler_step(s, a0, a1);
    const double ex = s[0] - a.gx, ey = s[1] - a.gy;
    const bool reached = sqrt(fma(ey, ey, ex * ex)) < 0.5;
    const bool coll = car_collides(s[0], s[1], s[2], maze, rows, cols);
    result[0] = a0;
    result[1] = a1;
    if (coll) {
      result[2] = 2.0;                                   // the state stays; the nominal sequence restarts from rest
      for (int j = 0; j < 2 * a.T; ++j) U[j] = 0.0;
      for (int j = 0; j < 6; ++j) result[8 + j] = state_io[j];
    } else {
      result[2] = reached ? 1.0 : 0.0;
      for (int j = 0; j < 6; ++j) state_io[j] = s[j];
      for (int j = 0; j < 6; ++j) result[8 + j] = s[j];
      for (int t = 0; t + 1 < a.T; ++t) { U[2 * t] = U[2 * t + 2]; U[2 * t + 1] = U[2 * t + 3]; }   // last control held
    }
  }
}

}  // namespace

extern "C" int32_t ditree_mppi_step(ditree_ctx* ctx, const ditree_mppi_params* p, double* state_io, double* U_io,
                                    const double* path_xy, int32_t P, const double* goal_xy, const double* noise,
                                    uint64_t counter, int32_t stages, double* costs, double* weights, int32_t* flags,
                                    double* sums, double* result, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!ctx->maze) return set_err(ctx, DITREE_E_STATE, "mppi_step: no maze uploaded");
  if (!p || !state_io || !U_io || !path_xy || !goal_xy || !costs || !result || !sums || p->T < 1 || p->T > MPPI_MAX_T ||
      p->K < 1 || P < 1 || P > MPPI_MAX_P || !(p->lambda > 0.0) || !(p->sigma[0] > 0.0) || !(p->sigma[1] > 0.0) ||
      p->window_back < 0 || p->window_fwd < 0 || (stages & ~DITREE_MPPI_ALL) != 0 || stages == 0 || p->k_offset < 0)
    return set_err(ctx, DITREE_E_ARG, "mppi_step: bad argument (1 <= T <= 64, 1 <= P <= 4096, lambda, sigma > 0, stages 1..31)");
  if (p->lanes != 0 && p->lanes != 1 && p->lanes != 2 && p->lanes != 4)
    return set_err(ctx, DITREE_E_ARG, "mppi_step: lanes must be 0, 1, 2 or 4");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MppiArgs a;
  a.T = p->T; a.K = p->K; a.P = P;
  a.lambda = p->lambda; a.s0 = p->sigma[0]; a.s1 = p->sigma[1];
  a.w_track = p->w_track; a.w_progress = p->w_progress; a.w_collision = p->w_collision; a.w_goal = p->w_goal;
  a.seed = p->seed; a.counter = counter;
  a.wback = p->window_back; a.wfwd = p->window_fwd;
  a.gx = goal_xy[0]; a.gy = goal_xy[1];
  a.k0 = p->k_offset;
  const int slices = std::min(MPPI_SLICES, (a.K + 255) / 256);      // 256 rollouts per slice up to 65 536, more beyond
  if (!ctx->mppi_partial || !ctx->mppi_minkey) {             // both or neither: a failed second allocation is retried, never skipped
    if (!ctx->mppi_partial)
      HIP_TRY(ctx, hipMalloc((void**)&ctx->mppi_partial, (size_t)MPPI_SLICES * (3 + 2 * MPPI_MAX_T) * sizeof(double)));
    if (!ctx->mppi_minkey) HIP_TRY(ctx, hipMalloc((void**)&ctx->mppi_minkey, sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->mppi_minkey, 0xFF, sizeof(unsigned long long), s));
  }
  if (stages & DITREE_MPPI_ROLLOUTS) {
    const size_t lds = (size_t)P * 16 + (size_t)a.T * 16 + 128 + (((size_t)ctx->rows * ctx->cols + 15) & ~(size_t)15);
    if (lds > 160 * 1024) return set_err(ctx, DITREE_E_ARG, "mppi_step: path + maze exceed the LDS");
    const int G = p->lanes == 0 ? MPPI_DEFAULT_LANES : p->lanes;
    const long long threads = (long long)a.K * G;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    const void* fn = G == 4 ? (const void*)mppi_rollout_kernel<4> : (G == 2 ? (const void*)mppi_rollout_kernel<2> : (const void*)mppi_rollout_kernel<1>);
    static bool attr_done[64][3] = {};
    if (lds > 64 * 1024 && ctx->device < 64 && !attr_done[ctx->device][G >> 1]) {
      HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[ctx->device][G >> 1] = true;
    }
#define MPPI_LAUNCH(GG)                                                                                                     \
  hipLaunchKernelGGL(mppi_rollout_kernel<GG>, grid, block, lds, s, ctx->maze, ctx->rows, ctx->cols, state_io, U_io,          \
                     (const double2*)path_xy, noise, a, costs, flags, result, ctx->mppi_minkey)
    HIP_TRY(ctx, hipMemsetAsync(ctx->mppi_minkey, 0xFF, sizeof(unsigned long long), s));
    if (G == 4) MPPI_LAUNCH(4);
    else if (G == 2) MPPI_LAUNCH(2);
    else MPPI_LAUNCH(1);
#undef MPPI_LAUNCH
  }
  if (stages & DITREE_MPPI_MIN) hipLaunchKernelGGL(mppi_min_kernel, dim3(1), dim3(1), 0, s, ctx->mppi_minkey, result);
  if (stages & DITREE_MPPI_SUMS)
    hipLaunchKernelGGL(mppi_partial_kernel, dim3(slices, a.T), dim3(256), 0, s, costs, noise, a, result, ctx->mppi_partial, weights, flags);
  if (stages & (DITREE_MPPI_SUMS | DITREE_MPPI_APPLY | DITREE_MPPI_EXECUTE))
    hipLaunchKernelGGL(mppi_finish_kernel, dim3(1), dim3(256), 0, s, ctx->maze, ctx->rows, ctx->cols, ctx->mppi_partial, slices,
                       a, state_io, U_io, weights, sums, result, (stages & DITREE_MPPI_SUMS) ? 1 : 0,
                       (stages & DITREE_MPPI_APPLY) ? 1 : 0, (stages & DITREE_MPPI_EXECUTE) ? 1 : 0);
  if ((stages & DITREE_MPPI_APPLY) && weights != nullptr)
    hipLaunchKernelGGL(mppi_weights_kernel, dim3((a.K + 255) / 256), dim3(256), 0, s, weights, sums, a.K);
  HIP_TRY(ctx, hipGetLastError());
  return DITREE_OK;
}
