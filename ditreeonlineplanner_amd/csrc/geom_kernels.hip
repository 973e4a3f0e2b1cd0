// Per-candidate kinodynamic kernels of the DiTree expansion path (gfx950, FP64).
//
// This translation unit is compiled with -ffp-contract=off: the reference computes
// every expression with separately rounded numpy operations, and flags / parent
// indices must be bit-exact against the CPU oracle, so no a*b+c may be fused here
// except where the reference's LAPACK does (lidar border solve, explicit fma()).
//
// Reference sites (paths relative to the reference root):
//   nn_argmin      planners/RRT.py:49-51 (KDTree.query k=1 on state[:2])
//   local_map      common/map_utils.py:391-459
//   cond_vector    policies/fm_policy.py:60-143 (car branch)
//   car_rollout    planners/base_planner.py:257-320, car_env.py:341-396,
//                  common/map_utils.py:103-115,221-329
//   lidar_scan     lidar_sim/lidar_2d_sim.py:18-98
//   accept/commit  planners/RRT.py:179-217
#include "car_device.h"
#include "ditree_internal.h"

#define WAVE 64

// ------------------------------------------------------------------------- maze
__global__ void maze_convert_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float v = src[i];
    int c = (int)v;
    dst[i] = (v == (float)c && c >= 0 && c < 256) ? (unsigned char)c : (unsigned char)255;
  }
}
void launch_maze_convert(const float* src, unsigned char* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(maze_convert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
}

// Stage the maze into LDS (cells are single bytes; every shipped maze is <= 31x31).
__device__ __forceinline__ void stage_maze(unsigned char* lds, const unsigned char* __restrict__ g, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
}

// ------------------------------------------------------------------------- nearest node
// One wave per group of QPW queries; lanes stride the node array (coalesced 16-B loads),
// then a 64-lane arg-min reduction carrying (distance, index); ties -> lowest index.
#define NN_QPW 4
__global__ void __launch_bounds__(256)
nn_argmin_kernel(const double* __restrict__ queries, int q_stride, int B, const double2* __restrict__ node_xy,
                 int N, int32_t* __restrict__ out_idx, const double* __restrict__ node_state,
                 const double* __restrict__ node_last_action, const uint8_t* __restrict__ node_has_prev,
                 double* __restrict__ out_state, double* __restrict__ out_prev_action,
                 uint8_t* __restrict__ out_has_prev, int S, int D) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int q0 = wave * NN_QPW;
  if (q0 >= B) return;
  double qx[NN_QPW], qy[NN_QPW], best[NN_QPW];
  int bidx[NN_QPW];
#pragma unroll
  for (int k = 0; k < NN_QPW; ++k) {
    int q = min(q0 + k, B - 1);
    qx[k] = queries[(size_t)q * q_stride + 0];
    qy[k] = queries[(size_t)q * q_stride + 1];
    best[k] = __builtin_huge_val();
    bidx[k] = 0x7fffffff;
  }
  // eight nodes per trip: their loads are independent (one load - wait - compare round per node exposes a memory round trip
  // N / 64 times: 1.1 ms of a round at 10^5 nodes), only the running minima chain; indices past N are clamped for the load and
  // masked for the compare, the scan order per lane is unchanged (increasing index, strict '<': lowest index wins)
  for (int i = lane; i < N; i += 8 * WAVE) {
    double2 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = node_xy[min(i + u * WAVE, N - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int iu = i + u * WAVE;
      if (iu < N) {
#pragma unroll
        for (int k = 0; k < NN_QPW; ++k) {
          double dx = qx[k] - p[u].x, dy = qy[k] - p[u].y;
          double d = dx * dx + dy * dy;
          if (d < best[k]) { best[k] = d; bidx[k] = iu; }     // strict: keeps the lowest index per lane
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NN_QPW; ++k) {
    double d = best[k];
    int ix = bidx[k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      double od = __shfl_xor(d, m);
      int oi = __shfl_xor(ix, m);
      if (od < d || (od == d && oi < ix)) { d = od; ix = oi; }
    }
    // NaN distances never win a '<'; an all-NaN scan leaves 0x7fffffff -> clamp to node 0
    if (ix == 0x7fffffff) ix = 0;
    int q = q0 + k;
    if (q < B) {
      if (lane == 0) out_idx[q] = ix;
      if (node_state != nullptr) {
        if (lane < S) out_state[(size_t)q * S + lane] = node_state[(size_t)ix * S + lane];               // S <= 64
        if (lane >= 32 && lane < 32 + D) out_prev_action[(size_t)q * D + (lane - 32)] = node_last_action[(size_t)ix * D + (lane - 32)];
        if (lane == 63) out_has_prev[q] = node_has_prev[ix];
      }
    }
  }
}
void launch_nn_argmin(const double* queries, int q_stride, int B, const double* node_xy, int N, int32_t* out_idx,
                      const double* node_state, const double* node_last_action, const uint8_t* node_has_prev,
                      double* out_state, double* out_prev_action, uint8_t* out_has_prev, hipStream_t s, int S, int D) {
  int waves = (B + NN_QPW - 1) / NN_QPW;
  int blocks = (waves + 3) / 4;
  hipLaunchKernelGGL(nn_argmin_kernel, dim3(blocks), dim3(256), 0, s, queries, q_stride, B,
                     (const double2*)node_xy, N, out_idx, node_state, node_last_action, node_has_prev, out_state,
                     out_prev_action, out_has_prev, S, D);
}

// ------------------------------------------------------------------------- local map
// One workgroup per candidate; maze staged in LDS; one thread per output cell.
__global__ void __launch_bounds__(256)
local_map_kernel(const unsigned char* __restrict__ maze, int rows, int cols, const double* __restrict__ state,
                 int state_stride, const int32_t* __restrict__ active, const int32_t* __restrict__ idx, int n, AxisArg axis,
                 double s_global, int scaled, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int ob = blockIdx.x;                                        // dense output row
  const int b = idx ? idx[ob] : ob;                                 // candidate (compacted rounds pass an index list)
  if (active != nullptr && active[b] != DITREE_ST_OK) return;      // block-uniform
  stage_maze(lds, maze, rows * cols);
  // RRT.py:158-166 passes curr_state[0], curr_state[1], curr_state[2] for every env (for the ant, element 2 is the torso
  // height, not a heading -- the reference's own behaviour, kept)
  const double x = state[(size_t)b * state_stride + 0], y = state[(size_t)b * state_stride + 1], th = state[(size_t)b * state_stride + 2];
  double c, sn;
  sincos(th, &sn, &c);          // one range reduction; ocml's sin / cos evaluate the same kernels (identical values)
  // base_planner.py:88-89,100-101,113-114: centre = (W/2, H/2) * maze_size_scaling (car: 1, ant: s_global = 4); RRT.py:166
  const double cx = (double)cols / 2.0 * s_global, cy = (double)rows / 2.0 * s_global;
  for (int cell = threadIdx.x; cell < n * n; cell += blockDim.x) {
    int i = cell / n, j = cell - i * n;                             // meshgrid: x_local[i][j] = xs[j], y_local = ys[i]
    double xl = axis.v[j], yl = axis.v[i];
    double xg = c * xl - sn * yl + x;                               // map_utils.py:443
    double yg = sn * xl + c * yl + y;                               // :444
    double fy = floor((cy - yg) / s_global);                        // :447
    double fx = floor((xg + cx) / s_global);                        // :448
    // np.clip after astype(int): NaN / huge values become INT64_MIN in numpy -> clip to 0
    int yi = (fy >= 0.0) ? ((fy < (double)rows) ? (int)fy : rows - 1) : 0;
    int xi = (fx >= 0.0) ? ((fx < (double)cols) ? (int)fx : cols - 1) : 0;
    float m = (float)lds[yi * cols + xi];
    out[(size_t)ob * n * n + cell] = scaled ? (m * 2.0f - 1.0f) : m;
  }
}
void launch_local_map(const unsigned char* maze, int rows, int cols, const double* state, const int32_t* active,
                      const int32_t* idx, int B, int n, const AxisArg& axis, double s_global, int scaled, float* out,
                      hipStream_t s, int state_stride) {
  size_t lds = ((size_t)rows * cols + 15) & ~(size_t)15;
  hipLaunchKernelGGL(local_map_kernel, dim3(B), dim3(256), lds, s, maze, rows, cols, state, state_stride, active, idx, n, axis,
                     s_global, scaled, out);
}

// ------------------------------------------------------------------------- conditioning vector
__global__ void cond_vector_kernel(const double* __restrict__ state, const double* __restrict__ prev_action,
                                   const uint8_t* __restrict__ has_prev, const double* __restrict__ cond_goal,
                                   const int32_t* __restrict__ idx, int B, NormArg nm, double lm_size,
                                   float* __restrict__ out) {
  const int ob = blockIdx.x * blockDim.x + threadIdx.x;
  if (ob >= B) return;
  const int b = idx ? idx[ob] : ob;
  const double* st = state + (size_t)b * 6;
  float* o = out + (size_t)ob * 7;
  // fm_policy.py:76,108,110,112: normalise in f64, drop x,y,psi, cast to f32
#pragma unroll
  for (int k = 0; k < 3; ++k) o[k] = (float)((st[3 + k] - nm.obs_mean[3 + k]) / nm.obs_std[3 + k]);
  // :113-123: zeros stay un-normalised when prev_actions is None
  if (has_prev[b]) {
    o[3] = (float)((prev_action[(size_t)b * 2 + 0] - nm.act_mean[0]) / nm.act_std[0]);
    o[4] = (float)((prev_action[(size_t)b * 2 + 1] - nm.act_mean[1]) / nm.act_std[1]);
  } else {
    o[3] = 0.0f;
    o[4] = 0.0f;
  }
  // :125-143 in f32: g = float(goal - pos); R(-yaw) g; tanh(g / lm_size)
  float gx = (float)(cond_goal[(size_t)b * 2 + 0] - st[0]);
  float gy = (float)(cond_goal[(size_t)b * 2 + 1] - st[1]);
  float yaw = (float)st[2];
  float c = cosf(yaw), sn = sinf(yaw);
  float rx = c * gx + sn * gy;
  float ry = (-sn) * gx + c * gy;
  float sc = (float)lm_size;
  o[5] = tanhf(rx / sc);
  o[6] = tanhf(ry / sc);
}
void launch_cond_vector(const double* state, const double* prev_action, const uint8_t* has_prev,
                        const double* cond_goal, const int32_t* idx, int B, const NormArg& nm, double lm_size,
                        float* out, hipStream_t s) {
  hipLaunchKernelGGL(cond_vector_kernel, dim3((B + 255) / 256), dim3(256), 0, s, state, prev_action, has_prev,
                     cond_goal, idx, B, nm, lm_size, out);
}

// policies/fm_policy.py:60-143, antmaze branch.  obs (B, n_hist, 29) f64: [x, y | 27 observations]; the 27 are normalised
// (:77), the quaternion is taken FROM THE NORMALISED values (:78, elements 3..6 = x, y, z, w) and replaced by the first two
// columns of its rotation matrix (common/se3_utils.py:177-189), the last `obs_history` = 3 steps are kept (zero rows in
// front when fewer are given, :96-102), x, y dropped (:107): 3 x 29 = 87 values; then the previous action (8, normalised;
// raw zeros when there is none, :113-123) and tanh((goal - position) / local_map_size) with yaw = 0 (:82,125-143).
// Two input forms: obs (B, n_rows, 29) with n_rows = n_hist given steps each (hist_n == NULL), or -- the round's buffers --
// obs (B, 3, 29) with the hist_n[b] valid steps at the END of the three rows (n_rows = 3).  idx: compacted rounds (dense output
// row ob <- candidate idx[ob]).
__global__ void cond_vector_ant_kernel(const double* __restrict__ obs, int n_rows, const int32_t* __restrict__ hist_n,
                                       const double* __restrict__ prev_action,
                                       const uint8_t* __restrict__ has_prev, const double* __restrict__ cond_goal,
                                       const int32_t* __restrict__ idx, int B,
                                       AntNormArg nm, double lm_size, float* __restrict__ out) {
  const int ob = blockIdx.x * blockDim.x + threadIdx.x;
  if (ob >= B) return;
  const int b = idx ? idx[ob] : ob;
  const int n_hist = hist_n ? hist_n[b] : n_rows;
  float* o = out + (size_t)ob * 97;
  for (int hs = 0; hs < 3; ++hs) {                        // slot hs of the 3-step history <- given step hs - (3 - n_hist)
    const int src = hs - (3 - n_hist) + (n_rows - n_hist);          // the valid steps are the LAST n_hist of the n_rows rows
    float* oh = o + hs * 29;
    if (hs < 3 - n_hist) {
      for (int k = 0; k < 29; ++k) oh[k] = 0.0f;
      continue;
    }
    const double* st = obs + ((size_t)b * n_rows + src) * 29;
    double v[27];
    for (int k = 0; k < 27; ++k) v[k] = (st[2 + k] - nm.obs_mean[k]) / nm.obs_std[k];
    // obs_seq[..., 3:7] = v[1..4] = (qx, qy, qz, qw)
    const double qx = v[1], qy = v[2], qz = v[3], qw = v[4];
    const double r00 = 1.0 - 2.0 * (qy * qy + qz * qz);
    const double r10 = 2.0 * (qx * qy + qw * qz);
    const double r20 = 2.0 * (qx * qz - qw * qy);
    const double r01 = 2.0 * (qx * qy - qw * qz);
    const double r11 = 1.0 - 2.0 * (qx * qx + qz * qz);
    const double r21 = 2.0 * (qy * qz + qw * qx);
    oh[0] = (float)v[0];                                  // torso height
    oh[1] = (float)r00; oh[2] = (float)r10; oh[3] = (float)r20; oh[4] = (float)r01; oh[5] = (float)r11; oh[6] = (float)r21;
    for (int k = 5; k < 27; ++k) oh[2 + k] = (float)v[k];  // obs_seq[..., 7:] -> 22 values
  }
  float* oa = o + 87;
  if (has_prev[b]) {
    for (int k = 0; k < 8; ++k) oa[k] = (float)((prev_action[(size_t)b * 8 + k] - nm.act_mean[k]) / nm.act_std[k]);
  } else {
    for (int k = 0; k < 8; ++k) oa[k] = 0.0f;
  }
  const double* last = obs + ((size_t)b * n_rows + (n_rows - 1)) * 29;        // position = obs_seq[:, -1, :2] (:74)
  const float gx = (float)(cond_goal[(size_t)b * 2 + 0] - last[0]);
  const float gy = (float)(cond_goal[(size_t)b * 2 + 1] - last[1]);
  const float sc = (float)lm_size;
  // yaw = 0: the rotation matrix [[1, 0], [-0, 1]] leaves g unchanged
  o[95] = tanhf(gx / sc);
  o[96] = tanhf(gy / sc);
}
void launch_cond_vector_ant(const double* obs, int n_rows, const int32_t* hist_n, const double* prev_action, const uint8_t* has_prev,
                            const double* cond_goal, const int32_t* idx, int B, const AntNormArg& nm, double lm_size, float* out,
                            hipStream_t s) {
  hipLaunchKernelGGL(cond_vector_ant_kernel, dim3((B + 127) / 128), dim3(128), 0, s, obs, n_rows, hist_n, prev_action, has_prev,
                     cond_goal, idx, B, nm, lm_size, out);
}

// ------------------------------------------------------------------------- reference path behind the obstacle
// planners/RRT.py:83-111 extract_path_after_obstacle: c = argmin_i ||cur - path_i|| (first occurrence, np.linalg.norm along
// axis 1 = sqrt(dx*dx + dy*dy)) in the dtype numpy gives `env.state[:2] - path[:, :2]` -- float64 when the env state is float64
// (after any env step; the f32 path is promoted), float32 right after env.reset (car_env.py:215 builds a float32 state) --;
// over rest = path[c:], cells by cell_xy_to_rowcol (the path's f32, floor); i_b = first rest index in an occupied cell (none: -1, the reference then indexes the LAST point); then the
// reference's while loop walks to the first free point j >= i_b and returns rest[j + 1:] (k = j + 1; the blocked run reaching
// the end gives k = len(rest); no crossing gives k = -1, i.e. the last point alone when that one is free).
// out[0] = c, out[1] = k.  One work-group; the path is a few hundred to a few thousand points.
__global__ void __launch_bounds__(256)
path_after_obstacle_kernel(const float* __restrict__ path, int stride, int P, double cx, double cy, int f32_state,
                           const unsigned char* __restrict__ maze, int rows, int cols, int32_t* __restrict__ out) {
  __shared__ double s_d[256];
  __shared__ int s_i[256];
  __shared__ int s_c, s_ib, s_j;
  const int tid = threadIdx.x;
  double best = __builtin_huge_val();
  int bi = 0x7fffffff;
  for (int i = tid; i < P; i += 256) {
    double d;
    if (f32_state) {
      const float dx = (float)cx - path[(size_t)i * stride], dy = (float)cy - path[(size_t)i * stride + 1];
      d = (double)sqrtf(dx * dx + dy * dy);
    } else {
      const double dx = cx - (double)path[(size_t)i * stride], dy = cy - (double)path[(size_t)i * stride + 1];
      d = sqrt(dx * dx + dy * dy);
    }
    if (d < best) { best = d; bi = i; }
  }
  s_d[tid] = best; s_i[tid] = bi;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) {
      const double od = s_d[tid + o];
      const int oi = s_i[tid + o];
      if (od < s_d[tid] || (od == s_d[tid] && oi < s_i[tid])) { s_d[tid] = od; s_i[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) { s_c = s_i[0]; s_ib = 0x7fffffff; s_j = 0x7fffffff; }
  __syncthreads();
  const int c = s_c, n = P - c;
  const float yc = (float)rows / 2.0f, xc = (float)cols / 2.0f;
  auto blocked = [&](int r) {                               // rest index r -> occupied cell?
    const float x = path[(size_t)(c + r) * stride], y = path[(size_t)(c + r) * stride + 1];
    const int i = (int)floorf((yc - y) / 1.0f), j = (int)floorf((x + xc) / 1.0f);
    // numpy would wrap a negative index and raise beyond the map: paths of a planner stay inside; clamp instead of faulting
    const int ii = min(max(i, 0), rows - 1), jj = min(max(j, 0), cols - 1);
    return maze[ii * cols + jj] == 1;
  };
  int mine = 0x7fffffff;
  for (int r = tid; r < n; r += 256)
    if (blocked(r)) { mine = r; break; }
  if (mine != 0x7fffffff) atomicMin(&s_ib, mine);
  __syncthreads();
  const int ib = s_ib;
  if (ib == 0x7fffffff) {                                   // no crossing: the reference looks at the last point
    if (tid == 0) { out[0] = c; out[1] = blocked(n - 1) ? n : -1; }
    return;
  }
  mine = 0x7fffffff;
  for (int r = ib + tid; r < n; r += 256)
    if (!blocked(r)) { mine = r; break; }
  if (mine != 0x7fffffff) atomicMin(&s_j, mine);
  __syncthreads();
  if (tid == 0) { out[0] = c; out[1] = s_j == 0x7fffffff ? n : s_j + 1; }
}
void launch_path_after_obstacle(const float* path, int stride, int P, double cx, double cy, int f32_state, const unsigned char* maze,
                                int rows, int cols, int32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(path_after_obstacle_kernel, dim3(1), dim3(256), 0, s, path, stride, P, cx, cy, f32_state, maze, rows, cols, out);
}

// ------------------------------------------------------------------------- rollout
// planners/base_planner.py:257-320 + car_env.py:240-282,341-396 + common/map_utils.py:103-115 for a batch of candidates.
// One lane per candidate: the A Euler steps with their goal + two-ball collision tests are a sequential FP64 chain of
// ~ 800 instructions per step (2 sincos, cos, tanh, 2 hypot, sqrt; round 3: 1 200).  Work-group size by batch (launcher): one wave
// per group spreads a small batch over many CUs (the round's 1 024 candidates: 16 CUs instead of 4), 256 threads once every
// SIMD has a wave anyway.
//   The kernel is bound by that FP64 chain with one wave per SIMD (dependent FP64 instructions, ~ 8 cycles each), not by HBM:
//   65 536 x 16 steps take 41.6 us with candidate-minor rows, 35.6 us without any row stores (round 3: 66.6 us); HBM-side
//   traffic 94.3 MB for 93.8 MB of algorithmic bytes (round 3: 214 MB).  What it took, each step measured on its own
//   (profiles/NOTES.md): lockstep stores of candidate-minor rows; the action rows staged per lane in LDS by
//   one burst of loads per 16 steps and an explicit vmcnt(0) in front of the step loop (vmcnt counts stores: any wait for a load
//   inside the loop also waited for the previous step's row stores); one sincos per sin / cos pair, the nearest-corner test,
//   branch-free cell tests, tanh through expm1 (car_device.h, fp64_device.h).  Measured and dropped: wave-cooperative LDS staging
//   with barriers (100 us), register prefetch of the next steps' actions, software-pipelined steps, a wave-specialised two-wave
//   form, work-groups of 128 threads (+ 27 %).
// G lanes per candidate (1 or 2).  G = 2: both lanes of a pair integrate the (identical) dynamics, lane g tests ball g and the
// pair ORs by one lane exchange, lane 0 stores the state rows and lane 1 the action rows -- twice the waves for the same batch
// (two per SIMD at 65 536 candidates), each with a shorter chain per step.  Same arithmetic, same results.
template <int G, bool LOCKSTEP, bool STAGE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))   // the action burst holds 64 registers: no spills
car_rollout_kernel(const unsigned char* __restrict__ maze, int rows, int cols, double* __restrict__ state_io,
                   const double* __restrict__ actions, int64_t act_stride, int32_t* __restrict__ status_io, int B,
                   int A, double gx, double gy, double* __restrict__ states_out, ditree_strides sl,
                   double* __restrict__ actions_out, ditree_strides al, int32_t* __restrict__ steps_out,
                   int64_t steps_stride, int32_t* __restrict__ chunks_run, double* __restrict__ prev_action_io,
                   uint8_t* __restrict__ has_prev_io, const int32_t* __restrict__ idx, int act_dense,
                   const int32_t* __restrict__ budget, int chunk_j, ChunkStrides cs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  stage_maze(lds, maze, rows * cols);
  const int ob = (blockIdx.x * blockDim.x + threadIdx.x) / G, g = threadIdx.x & (G - 1);
  if (ob >= B) return;
  const int b = idx ? idx[ob] : ob;                 // compacted rounds: actions are dense (row ob), the rest per candidate
  if (status_io[b] != DITREE_ST_OK) return;
  // chunk_j < 0 (pool-scheduled early-exit rounds: one launch holds candidates at DIFFERENT chunks of their edges): the chunk
  // is the number of chunks this candidate has finished; the chunk-0 bases are offset by it
  if (chunk_j < 0) {
    chunk_j = chunks_run[b];
    if (states_out) states_out += (size_t)chunk_j * cs.states;
    if (actions_out) actions_out += (size_t)chunk_j * cs.actions_out;
    if (steps_out) steps_out += chunk_j;
    if (!act_dense) actions += (size_t)chunk_j * cs.actions_in;
  }
  if (budget != nullptr && chunk_j >= budget[b]) return;      // this visit's edge is shorter (prop_duration schedule)
  double s[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) s[k] = state_io[(size_t)b * 6 + k];
  const double* act = actions + (size_t)(act_dense ? ob : b) * act_stride;
  // STAGE: a lane copies sixteen steps of ITS actions into its LDS slots in ONE burst of loads (issued back to back, one wait) and
  // reads them back step by step -- no barrier (a lane only reads what it wrote); slot q of lane t at (q * blockDim + t) * 16: no
  // bank conflicts.  Two things are wrong with a 16-byte global load per step: the same 128-byte line is touched eight times up to
  // 4 us apart (re-fetched when the store stream has evicted it: 29 ... 114 MB per launch for 20 MB), and -- worse -- vmcnt retires
  // in order and counts stores, so waiting for step i's load also waits for the row stores of step i - 1: 45 % of the wave
  // cycles were s_waitcnt (profiles/r04_rollout_pmc_sq.json).  With the burst the step loop holds no global load at all.
  double2* abuf = STAGE ? reinterpret_cast<double2*>(lds + (((size_t)rows * cols + 15) & ~(size_t)15)) + threadIdx.x : nullptr;
  const int bd = blockDim.x;
  auto action_at = [&](int i, double& a0, double& a1) {
    if constexpr (STAGE) {
      if ((i & 15) == 0) {
        // sixteen named values, not an array: the scheduling barrier below is opaque to the optimiser, an array on its two sides
        // would stay in scratch memory
#define ACT_LD(q) const double2 t##q = *reinterpret_cast<const double2*>(act + 2 * min(i + q, A - 1))
        ACT_LD(0); ACT_LD(1); ACT_LD(2); ACT_LD(3); ACT_LD(4); ACT_LD(5); ACT_LD(6); ACT_LD(7);
        ACT_LD(8); ACT_LD(9); ACT_LD(10); ACT_LD(11); ACT_LD(12); ACT_LD(13); ACT_LD(14); ACT_LD(15);
#undef ACT_LD
        __builtin_amdgcn_sched_barrier(0);                        // all sixteen loads in flight before the first LDS write waits
#define ACT_ST(q) abuf[q * bd] = t##q
        ACT_ST(0); ACT_ST(1); ACT_ST(2); ACT_ST(3); ACT_ST(4); ACT_ST(5); ACT_ST(6); ACT_ST(7);
        ACT_ST(8); ACT_ST(9); ACT_ST(10); ACT_ST(11); ACT_ST(12); ACT_ST(13); ACT_ST(14); ACT_ST(15);
#undef ACT_ST
      }
      const double2 v = abuf[(i & 15) * bd];
      a0 = v.x; a1 = v.y;
    } else {
      a0 = act[2 * i]; a1 = act[2 * i + 1];
    }
  };
  // G = 2: lane 0 owns the state rows, lane 1 the action rows
  // row i, component k of a candidate's block: base + i * row + k * comp (packed rows: {6, 1}; step-major SoA: {6 B, B} --
  // then the 64 lanes of a wave store 512 contiguous bytes)
  double* so = (states_out && (G == 1 || g == 0)) ? states_out + (size_t)b * sl.cand : nullptr;
  double* ao = (actions_out && (G == 1 || g == 1)) ? actions_out + (size_t)b * al.cand : nullptr;
  if (so) {
#pragma unroll
    for (int k = 0; k < 6; ++k) so[k * sl.comp] = s[k];                         // states_sequence[0] = state
  }
  int status = DITREE_ST_OK;
  int steps = 0;
  double la0 = 0.0, la1 = 0.0;
  int i = 0;
  // Everything loaded so far (state, status, indices) has arrived BEFORE the step loop.  Left to the compiler, the wait for these
  // loads sits at their first use inside the loop, as vmcnt(0) -- and vmcnt counts the row stores of the previous step too: every
  // step then waited for its predecessor's stores to reach memory.
  __builtin_amdgcn_s_waitcnt(0x0F70);                                            // vmcnt(0)
  if constexpr (LOCKSTEP) {
    // Candidate-minor rows (sl.cand == 1) are stored in LOCKSTEP: every lane of the wave stores row i + 1 in the same
    // instruction -- its state while its edge is running, the zero row (base_planner.py:282) once it has ended -- so every store
    // is a whole 512-byte run.  (A lane that zero-filled its tail later, on its own, wrote partial sectors that the memory
    // system first had to fetch: 162 MB per launch for 94 MB of algorithmic bytes at 65 536 x 16, 111 MB in lockstep --
    // profiles/r04_rollout_*pmc_traffic.json.)
    bool alive = true;
    for (i = 0; i < A; ++i) {
      double c0r, c1r;
      action_at(i, c0r, c1r);
      if (alive) {
        car_euler_step(s, c0r, c1r);
        steps = i + 1;
        la0 = c0r; la1 = c1r;
      } else if (status == DITREE_ST_GOAL) {
        c0r = 0.0; c1r = 0.0;                                                     // :314-317; a collided edge's tail is copied through
      }
      if (so) {
#pragma unroll
        for (int k = 0; k < 6; ++k) so[(size_t)(i + 1) * sl.row + k * sl.comp] = alive ? s[k] : 0.0;
      }
      if (ao) { ao[(size_t)i * al.row] = c0r; ao[(size_t)i * al.row + al.comp] = c1r; }
      bool coll = false, done = false;
      if (alive) {
        const double ex = s[0] - gx, ey = s[1] - gy;
        done = sqrt(fma(ey, ey, ex * ex)) < 0.5;    // np.linalg.norm (ddot rounds as one fma), car_env.py:346-351
      }
      if constexpr (G == 2) {                       // both lanes of a pair stay in step (the exchange below is pair-wide)
        int mine = 0;
        if (alive) {
          const double off = 0.15 * 0.5, sgn = g ? -1.0 : 1.0;                   // common/map_utils.py:103-115: lane g, ball g
          double sps, cps;
          sincos(s[2], &sps, &cps);
          const double ox = off * cps, oy = off * sps;
          mine = ball_collides(s[0] + sgn * ox, s[1] + sgn * oy, lds, rows, cols) ? 1 : 0;
        }
        coll = (mine | __shfl_xor(mine, 1)) != 0;
      } else if (alive) {
        coll = car_collides(s[0], s[1], s[2], lds, rows, cols);                   // base_planner.py:306
      }
      if (alive && coll) {
        status = DITREE_ST_COLLIDED | (done ? DITREE_ST_FLAG_GOAL_AT_COLLISION : 0);
        alive = false;
      } else if (alive && done) {                                                 // :314-317
        status = DITREE_ST_GOAL;
        alive = false;
      }
    }
  } else {
    for (; i < A; ++i) {
      double a0r, a1r;
      action_at(i, a0r, a1r);
      car_euler_step(s, a0r, a1r);
      steps = i + 1;
      if (so) {
#pragma unroll
        for (int k = 0; k < 6; ++k) so[(size_t)(i + 1) * sl.row + k * sl.comp] = s[k];
      }
      if (ao) { ao[(size_t)i * al.row] = a0r; ao[(size_t)i * al.row + al.comp] = a1r; }
      la0 = a0r; la1 = a1r;
      double ex = s[0] - gx, ey = s[1] - gy;
      bool done = sqrt(fma(ey, ey, ex * ex)) < 0.5;   // np.linalg.norm (ddot rounds as one fma), car_env.py:346-351
      bool coll;                                                                   // base_planner.py:306
      if constexpr (G == 2) {
        const double off = 0.15 * 0.5, sgn = g ? -1.0 : 1.0;                       // common/map_utils.py:103-115: lane g, ball g
        double sps, cps;
        sincos(s[2], &sps, &cps);
        const double ox = off * cps, oy = off * sps;
        const int mine = ball_collides(s[0] + sgn * ox, s[1] + sgn * oy, lds, rows, cols) ? 1 : 0;
        coll = (mine | __shfl_xor(mine, 1)) != 0;
      } else {
        coll = car_collides(s[0], s[1], s[2], lds, rows, cols);
      }
      if (coll) {
        status = DITREE_ST_COLLIDED | (done ? DITREE_ST_FLAG_GOAL_AT_COLLISION : 0);
        ++i;
        break;
      }
      if (done) {                                                                   // :314-317
        status = DITREE_ST_GOAL;
        ++i;
        break;
      }
    }
    // rows after the last executed step stay zero (states :282; actions zeroed :315)
    for (int r = i; r < A; ++r) {
      if (so) {
#pragma unroll
        for (int k = 0; k < 6; ++k) so[(size_t)(r + 1) * sl.row + k * sl.comp] = 0.0;
      }
      if (ao) {
        // only the goal branch zeroes the remaining actions (:314-317); a collided edge is discarded
        // by the caller, its untouched tail is copied through like the reference's array
        const bool z = (status == DITREE_ST_GOAL);
        ao[(size_t)r * al.row] = z ? 0.0 : act[2 * r];
        ao[(size_t)r * al.row + al.comp] = z ? 0.0 : act[2 * r + 1];
      }
    }
  }
  if (G == 2 && g != 0) return;                       // per-candidate results: lane 0
#pragma unroll
  for (int k = 0; k < 6; ++k) state_io[(size_t)b * 6 + k] = s[k];
  status_io[b] = status;
  if (steps_out) steps_out[(size_t)b * steps_stride] = steps;
  if (chunks_run) chunks_run[b] += 1;
  if (status == DITREE_ST_OK && prev_action_io) {                                 // RRT.py:188
    prev_action_io[(size_t)b * 2 + 0] = la0;
    prev_action_io[(size_t)b * 2 + 1] = la1;
    if (has_prev_io) has_prev_io[b] = 1;
  }
}
void launch_car_rollout_ex(const unsigned char* maze, int rows, int cols, double* state_io, const double* actions,
                           int64_t act_stride, int32_t* status_io, int B, int A, double gx, double gy,
                           double* states_out, ditree_strides states_stride, double* actions_out, ditree_strides actout_stride,
                           int32_t* steps_out, int64_t steps_stride, int32_t* chunks_run, double* prev_action_io,
                           uint8_t* has_prev_io, const int32_t* idx, int act_dense, hipStream_t s, const int32_t* budget,
                           int chunk_j, ChunkStrides cs) {
  size_t lds = ((size_t)rows * cols + 15) & ~(size_t)15;
  // lane-private LDS staging of the action rows (DITREE_ROLLOUT_STAGE=0 turns it off; needs 16-byte aligned rows): 65 536 x 16,
  // candidate-minor rows: 114 -> 20.4 MB fetched per launch (total 188 -> 94.3 MB = 1.005 x the algorithmic 93.8 MB), and no
  // global load or vmcnt wait left inside the step loop (profiles/r04_rollout_stage_* vs r04_rollout_nostage_*, NOTES.md)
  static int stage_env = -1;
  if (stage_env < 0) { const char* e = getenv("DITREE_ROLLOUT_STAGE"); stage_env = e ? atoi(e) : 1; }
  int stage = stage_env && (act_stride % 2 == 0) && (cs.actions_in % 2 == 0) && ((uintptr_t)actions % 16 == 0) ? 1 : 0;
  // work-group size: one wave per group spreads a small batch over many CUs (1 024 candidates: 16 CUs instead of 4); once
  // every SIMD has a wave anyway, four waves per group share one staged maze.  NOT two: 65 536 x 16 steps take 64.0 us in groups
  // of 128 threads against 50.5 (256) and 51.2 (64) -- profiles/r04_rollout_blocksize_probe.json; two-wave groups do not spread
  // evenly over the four SIMDs of a CU.  DITREE_ROLLOUT_BLK = 64 | 128 | 256 overrides.
  static int blk_env = -1;
  if (blk_env < 0) { const char* e = getenv("DITREE_ROLLOUT_BLK"); blk_env = e ? atoi(e) : 0; }
  const int blk = (blk_env == 64 || blk_env == 128 || blk_env == 256) ? blk_env : (B >= 16384 ? 256 : 64);
  static bool attr_done_dev[64] = {};                        // the attribute is per kernel AND per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool attr_done = dev >= 0 && dev < 64 && attr_done_dev[dev];
  if (!attr_done) {                                          // 256 threads x 256 B of staged actions + the maze exceed 64 KB
    const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
    (void)hipFuncSetAttribute((const void*)car_rollout_kernel<1, true, true>, at, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)car_rollout_kernel<1, false, true>, at, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)car_rollout_kernel<2, true, true>, at, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)car_rollout_kernel<2, false, true>, at, 160 * 1024);
    if (dev >= 0 && dev < 64) attr_done_dev[dev] = true;
  }
  if (lds + (size_t)blk * 256 > 160 * 1024) stage = 0;        // a maze that leaves no room for the staged actions: per-step loads
  // lanes per candidate, measured on MI355X (profiles/r03_rollout_lanes.json): 8 192 candidates x 8 steps 31.6 -> 26.2 us with
  // two lanes (the batch fills a quarter of the SIMDs: the second lane's wave is free), 65 536 x 16 steps 73.2 -> 80.0 us
  // (every SIMD already has a wave; the duplicated dynamics cost more than the shorter chain saves).  DITREE_ROLLOUT_LANES
  // = 1 | 2 overrides.  (A wave-specialised form -- two waves per candidate set, one per half of the step's chain, two barriers
  // per step -- gave the same results and the same 56 us: profiles/NOTES.md.)
  static int lanes_env = -1;
  if (lanes_env < 0) { const char* e = getenv("DITREE_ROLLOUT_LANES"); lanes_env = e ? atoi(e) : 0; }
  // lockstep row stores pay with candidate-minor rows (whole 512-byte runs); with rows packed per candidate they cost 27 %
  // (dead lanes keep storing zero rows step by step: profiles/r04_rollout_layout_probe.json) -- chosen by the layout
  const bool lock = (states_out && states_stride.cand == 1) || (!states_out && actions_out && actout_stride.cand == 1);
  const int G = lanes_env == 1 ? 1 : (lanes_env == 2 ? 2 : (B <= 32768 ? 2 : 1));
#define CAR_ROLLOUT_LAUNCH(GG, PP, SS)                                                                                          \
  hipLaunchKernelGGL((car_rollout_kernel<GG, PP, SS>), dim3((GG * B + blk - 1) / blk), dim3(blk), lds + (SS ? (size_t)blk * 256 : 0), s, maze, \
                     rows, cols, state_io, actions, act_stride, status_io, B, A, gx, gy, states_out, states_stride, actions_out,  \
                     actout_stride, steps_out, steps_stride, chunks_run, prev_action_io, has_prev_io, idx, act_dense, budget, chunk_j, cs)
#define CAR_ROLLOUT_PICK(GG)                                                                        \
  do {                                                                                              \
    if (lock) { if (stage) CAR_ROLLOUT_LAUNCH(GG, true, true); else CAR_ROLLOUT_LAUNCH(GG, true, false); }   \
    else { if (stage) CAR_ROLLOUT_LAUNCH(GG, false, true); else CAR_ROLLOUT_LAUNCH(GG, false, false); }      \
  } while (0)
  if (G == 2) CAR_ROLLOUT_PICK(2); else CAR_ROLLOUT_PICK(1);
#undef CAR_ROLLOUT_PICK
#undef CAR_ROLLOUT_LAUNCH
}
void launch_car_rollout(const unsigned char* maze, int rows, int cols, double* state_io, const double* actions,
                        int64_t act_stride, int32_t* status_io, int B, int A, double gx, double gy,
                        double* states_out, int64_t states_stride, double* actions_out, int64_t actout_stride,
                        int32_t* steps_out, double* prev_action_io, uint8_t* has_prev_io, hipStream_t s) {
  launch_car_rollout_ex(maze, rows, cols, state_io, actions, act_stride, status_io, B, A, gx, gy, states_out,
                        ditree_strides{states_stride, 6, 1}, actions_out, ditree_strides{actout_stride, 2, 1}, steps_out, 1,
                        nullptr, prev_action_io, has_prev_io, nullptr, 1, s, nullptr, 0, ChunkStrides{0, 0, 0});
}

// ------------------------------------------------------------------------- lidar
// One workgroup per pose, one thread per ray (181 rays -> 192 threads).  The maze (f32
// {0,1}) is staged into LDS as bytes; visited cells are OR-ed into an LDS bitmap.
__device__ __forceinline__ bool border_solve(double rx, double ry, double dx, double dy, double b0, double b1,
                                             double* t, double* s) {
  // np.linalg.solve([[rx, -dx], [ry, -dy]], b) as OpenBLAS evaluates it (verified bit-exact on
  // the build host): partial pivoting, multiplier scaled by the reciprocal pivot, FMA updates,
  // true divisions in the back substitution.  lidar_2d_sim.py:69-77.
  double a00 = rx, a01 = -dx, a10 = ry, a11 = -dy;
  if (fabs(a10) > fabs(a00)) {
    double t0 = a00; a00 = a10; a10 = t0;
    t0 = a01; a01 = a11; a11 = t0;
    t0 = b0; b0 = b1; b1 = t0;
  }
  if (a00 == 0.0) return false;
  double l = a10 * (1.0 / a00);
  double u11 = fma(-l, a01, a11);
  if (u11 == 0.0) return false;                          // LinAlgError: singular (ray parallel to border)
  double y1 = fma(-l, b0, b1);
  double x1 = y1 / u11;
  double x0 = fma(-a01, x1, b0) / a00;
  *t = x0;
  *s = x1;
  return true;
}

// One ray of Lidar2DSim._cast_ray + the endpoint re-derivation of scan() (lidar_2d_sim.py:18-98).  `mz` holds
// 1 for occupied cells, visited cells are OR-ed into `vis`.
__device__ __forceinline__ void lidar_ray(double x0, double y0, double yaw, int ray, const unsigned char* mz, int rows,
                                          int cols, unsigned char* vis, double* d_out, double* ex_out, double* ey_out,
                                          bool* hit_out) {
  // maze_width, maze_height = maze_data.shape (sic, :51): "width" = rows
  const double mw = (double)rows, mh = (double)cols;
  const double angle = -180.0 + 2.0 * (double)ray;                 // np.arange(-180, 182, 2)
  const double ang = (yaw + angle) * (M_PI / 180.0);               // np.deg2rad(yaw + angle) (:53-54)
  double rx, ry;
  sincos(ang, &ry, &rx);
  // borders Left, Right, Bottom, Top (:57-62): origin b0, direction d
  const double bx[4] = {0.0, mw, 0.0, 0.0}, by[4] = {0.0, 0.0, 0.0, mh};
  const double dxs[4] = {0.0, 0.0, mw, mw}, dys[4] = {mh, mh, 0.0, 0.0};
  double lx = x0, ly = y0;
  bool found = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (found) continue;
    double t, s;
    if (border_solve(rx, ry, dxs[k], dys[k], bx[k] - x0, by[k] - y0, &t, &s)) {
      if (t >= 0.0 && s <= 1.0 && s >= 0.0) {                        // :74
        lx = t * rx + x0;
        ly = t * ry + y0;
        found = true;
      }
    }
  }
  const double vx = lx - x0, vy = ly - y0;
  const double len = sqrt(fma(vy, vy, vx * vx));                     // np.linalg.norm (ddot = one fma)
  const double step = 0.1 / len;                                     // :85
  // len(np.arange(0, 1, step)) = ceil((1 - 0) / step)
  double nf = ceil(1.0 / step);
  long long n = (found && nf > 0.0 && nf < 1e9) ? (long long)nf : 0;
  bool h = false;
  double ox = lx, oy = ly;
  for (long long i = 0; i < n; ++i) {
    double t = (double)i * step;                                     // arange value start + i*delta
    double px = x0 + t * vx, py = y0 + t * vy;                       // :86
    double fx = floor(px), fy = floor(py);
    int qx = fx >= 0.0 ? (fx < mw ? (int)fx : rows - 1) : 0;         // clip to [0, maze_width-1]  (:89)
    int qy = fy >= 0.0 ? (fy < mh ? (int)fy : cols - 1) : 0;         // clip to [0, maze_height-1]
    // maze_data[q_y, q_x] (:91): q_y indexes rows.  On non-square maps the reference's swapped
    // clip bounds can index past the array (IndexError); clamp so the device never faults.
    int rr = min(qy, rows - 1), cc = min(qx, cols - 1);
    if (mz[rr * cols + cc] == 1) {
      h = true;
      ox = px;
      oy = py;
      break;
    }
    vis[rr * cols + cc] = 1;                                         // benign race: all writers store 1
  }
  double ex = ox - x0, ey = oy - y0;
  double d = sqrt(fma(ey, ey, ex * ex));                             // :96
  d = d < 0.0 ? 0.0 : (d > 300.0 ? 300.0 : d);                       // scan(): np.clip(d, 0, max_range) (:33)
  *d_out = d;
  *ex_out = x0 + d * rx;                                             // :36-39
  *ey_out = y0 + d * ry;
  *hit_out = h;
}

__global__ void __launch_bounds__(192)
lidar_scan_kernel(const double* __restrict__ poses, const float* __restrict__ maze, int rows, int cols,
                  double* __restrict__ dist, double* __restrict__ endpoints, uint8_t* __restrict__ hit,
                  uint8_t* __restrict__ visited) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int ncell = rows * cols;
  unsigned char* mz = lds;
  unsigned char* vis = lds + ((ncell + 15) & ~15);
  for (int i = threadIdx.x; i < ncell; i += blockDim.x) {
    mz[i] = (maze[i] == 1.0f) ? 1 : 0;
    vis[i] = 0;
  }
  __syncthreads();
  const int b = blockIdx.x;
  const int ray = threadIdx.x;
  const double x0 = poses[(size_t)b * 3 + 0], y0 = poses[(size_t)b * 3 + 1], yaw = poses[(size_t)b * 3 + 2];
  if (ray < DITREE_LIDAR_RAYS) {
    double d, ex, ey;
    bool h;
    lidar_ray(x0, y0, yaw, ray, mz, rows, cols, vis, &d, &ex, &ey, &h);
    dist[(size_t)b * DITREE_LIDAR_RAYS + ray] = d;
    endpoints[((size_t)b * DITREE_LIDAR_RAYS + ray) * 2 + 0] = ex;
    endpoints[((size_t)b * DITREE_LIDAR_RAYS + ray) * 2 + 1] = ey;
    hit[(size_t)b * DITREE_LIDAR_RAYS + ray] = h ? 1 : 0;
  }
  if (visited != nullptr) {
    __syncthreads();
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) visited[(size_t)b * ncell + i] = vis[i];
  }
}
void launch_lidar_scan(const double* poses, int B, const float* maze, int rows, int cols, double* dist,
                       double* endpoints, uint8_t* hit, uint8_t* visited, hipStream_t s) {
  size_t n = ((size_t)rows * cols + 15) & ~(size_t)15;
  hipLaunchKernelGGL(lidar_scan_kernel, dim3(B), dim3(192), 2 * n, s, poses, maze, rows, cols, dist, endpoints,
                     hit, visited);
}

// ------------------------------------------------------------------------- plan following (online loop)
// run_scenarios_with_lidar_DiTree.py:470-506 (run_type < 4) as ONE launch: execute the plan's actions one env step
// at a time (propagate_action_sequence_env: step, goal test, collision test against the KNOWN maze), and every
// time the accumulated step time exceeds scan_time scan the TRUE maze from the pose (scan_and_update_maze,
// :112-127), write the ray end cells into the known / scanned mazes and test the planned path for a crossing
// (check_no_obstacles_in_path, :158-181).  One workgroup: thread 0 integrates, threads 0..180 cast the rays,
// all threads scan the path.  The three mazes live in LDS for the whole launch.
__device__ __forceinline__ unsigned char maze_code(float v) {
  int c = (int)v;
  return (v == (float)c && c >= 0 && c < 256) ? (unsigned char)c : (unsigned char)255;
}
__global__ void __launch_bounds__(256)
follow_plan_kernel(double* __restrict__ state_io, const float* __restrict__ actions, int n_actions, int action_idx,
                   const float* __restrict__ path, int P, float* __restrict__ known, const float* __restrict__ truth,
                   float* __restrict__ scanned, unsigned char* __restrict__ known_codes, int rows, int cols, double gx,
                   double gy, double dt, double scan_time, double* __restrict__ executed, int32_t* __restrict__ result) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int ncell = rows * cols, pad = (ncell + 15) & ~15;
  unsigned char* kn = lds;                 // known maze codes
  unsigned char* tr = lds + pad;           // true maze: 1 = occupied
  unsigned char* sc = lds + 2 * pad;       // scanned maze codes
  unsigned char* vis = lds + 3 * pad;      // cells visited by the current scan
  unsigned char* dirty = lds + 4 * pad;    // cells of known / scanned written by this launch
  __shared__ double s[6];
  __shared__ int sh_event, sh_obstacle, sh_scan, sh_done;
  const int tid = threadIdx.x;
  for (int i = tid; i < ncell; i += blockDim.x) {
    kn[i] = maze_code(known[i]);
    tr[i] = truth[i] == 1.0f ? 1 : 0;
    sc[i] = maze_code(scanned[i]);
    vis[i] = 0;
    dirty[i] = 0;
  }
  if (tid < 6) s[tid] = state_io[tid];
  if (tid == 0) { sh_event = -1; sh_obstacle = 0x7fffffff; sh_scan = 0; sh_done = 0; }
  __syncthreads();
  double t_acc = 0.0;                      // thread 0 only
  int idx = action_idx, n_scans = 0;
  const float hf = (float)((double)rows / 2.0), wf = (float)((double)cols / 2.0);
  while (idx < n_actions) {
    if (tid == 0) {
      double st[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) st[k] = s[k];
      car_euler_step(st, (double)actions[2 * idx], (double)actions[2 * idx + 1]);
      const double ex = st[0] - gx, ey = st[1] - gy;
      const bool done = sqrt(fma(ey, ey, ex * ex)) < 0.5;                        // car_env.py:346-351
      if (car_collides(st[0], st[1], st[2], kn, rows, cols)) {                   // base_planner.py:306 -> done is None
        sh_event = 2;
      } else {
        double* eo = executed + (size_t)(idx - action_idx) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) { s[k] = st[k]; eo[k] = st[k]; }
        sh_done = done ? 1 : 0;
        t_acc += dt;
        sh_scan = t_acc > scan_time ? 1 : 0;
        if (sh_scan) t_acc = 0.0;
      }
    }
    __syncthreads();
    if (sh_event == 2) break;
    ++idx;
    if (sh_scan) {
      ++n_scans;
      // pose in fractional cell units (col, row), yaw in radians (:113-116)
      const double px = (s[0] + (double)cols / 2.0) / 1.0, py = ((double)rows / 2.0 - s[1]) / 1.0, yaw = s[2];
      double d, ex = 0.0, ey = 0.0;
      bool h;
      if (tid < DITREE_LIDAR_RAYS) lidar_ray(px, py, yaw, tid, tr, rows, cols, vis, &d, &ex, &ey, &h);
      __syncthreads();
      for (int i = tid; i < ncell; i += blockDim.x) {                            // scanned[visited] = 2 (:121)
        if (vis[i]) { sc[i] = 2; dirty[i] = 1; vis[i] = 0; }
      }
      __syncthreads();
      if (tid < DITREE_LIDAR_RAYS) {                                             // end cells -> occupied (:119-122)
        const double fx = floor(ex), fy = floor(ey);
        if (fx >= 0.0 && fx < (double)cols && fy >= 0.0 && fy < (double)rows) {
          const int c = (int)fy * cols + (int)fx;
          kn[c] = 1;
          sc[c] = 1;
          dirty[c] = 1;
        }
      }
      __syncthreads();
      int first = 0x7fffffff;                                                    // :158-181 in the path's float32
      for (int i = tid; i < P; i += blockDim.x) {
        const float row = (hf - path[2 * i + 1]) / 1.0f, col = (path[2 * i] + wf) / 1.0f;
        const float fr = floorf(row), fc = floorf(col);
        if (fr >= 0.0f && fr < (float)rows && fc >= 0.0f && fc < (float)cols && sc[(int)fr * cols + (int)fc] == 1) {
          first = i;
          break;
        }
      }
      if (first != 0x7fffffff) atomicMin(&sh_obstacle, first);
      __syncthreads();
    }
    if (sh_done) { if (tid == 0) sh_event = 1; break; }
    if (sh_obstacle != 0x7fffffff) { if (tid == 0) sh_event = 3; break; }
    __syncthreads();                       // thread 0 rewrites sh_scan / s in the next iteration
  }
  __syncthreads();
  for (int i = tid; i < ncell; i += blockDim.x) {
    if (dirty[i]) {
      known[i] = (float)kn[i];
      scanned[i] = (float)sc[i];
    }
    known_codes[i] = kn[i];
  }
  if (tid < 6) state_io[tid] = s[tid];
  if (tid == 0) {
    result[0] = sh_event < 0 ? 0 : sh_event;
    result[1] = idx;
    result[2] = sh_obstacle == 0x7fffffff ? -1 : sh_obstacle;
    result[3] = n_scans;
  }
}
void launch_follow_plan(double* state_io, const float* actions, int n_actions, int action_idx, const float* path, int P,
                        float* known, const float* truth, float* scanned, unsigned char* known_codes, int rows, int cols,
                        double gx, double gy, double dt, double scan_time, double* executed, int32_t* result,
                        hipStream_t s) {
  size_t pad = ((size_t)rows * cols + 15) & ~(size_t)15;
  hipLaunchKernelGGL(follow_plan_kernel, dim3(1), dim3(256), 5 * pad, s, state_io, actions, n_actions, action_idx, path,
                     P, known, truth, scanned, known_codes, rows, cols, gx, gy, dt, scan_time, executed, result);
}

// ------------------------------------------------------------------------- round bookkeeping
__global__ void round_begin_kernel(int32_t* status, int32_t* chunks_run, int32_t* chunk_steps, int B, int n_chunks) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  status[b] = DITREE_ST_OK;
  chunks_run[b] = 0;
  for (int j = 0; j < n_chunks; ++j) chunk_steps[(size_t)b * n_chunks + j] = 0;
}
void launch_round_begin(int32_t* status, int32_t* chunks_run, int32_t* chunk_steps, int B, int n_chunks,
                        hipStream_t s) {
  hipLaunchKernelGGL(round_begin_kernel, dim3((B + 255) / 256), dim3(256), 0, s, status, chunks_run, chunk_steps, B,
                     n_chunks);
}

// ------------------------------------------------------------------------- alive compaction
// idx_out[0..n) = candidates whose status is still DITREE_ST_OK, in candidate order; *count = n.
// One workgroup, ordered prefix sum (the denoiser then runs on the n alive rows only).
__global__ void __launch_bounds__(1024) compact_alive_kernel(const int32_t* __restrict__ status, int B,
                                                             int32_t* __restrict__ idx_out, int32_t* __restrict__ count,
                                                             const int32_t* __restrict__ budget, int next_chunk) {
  __shared__ int s_wave_sum[16];
  const int tid = threadIdx.x;
  int base = 0;
  for (int start = 0; start < B; start += blockDim.x) {
    const int b = start + tid;
    const int alive = (b < B && status[b] == DITREE_ST_OK && (budget == nullptr || next_chunk < budget[b])) ? 1 : 0;
    int v = alive;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int o = __shfl_up(v, d);
      if ((tid & 63) >= d) v += o;
    }
    if ((tid & 63) == 63) s_wave_sum[tid >> 6] = v;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      if (w < (tid >> 6)) woff += s_wave_sum[w];
      total += s_wave_sum[w];
    }
    if (alive) idx_out[base + woff + v - 1] = b;
    base += total;
    __syncthreads();
  }
  if (tid == 0) *count = base;
}
void launch_compact_alive(const int32_t* status, int B, int32_t* idx_out, int32_t* count, hipStream_t s,
                          const int32_t* budget, int next_chunk) {
  hipLaunchKernelGGL(compact_alive_kernel, dim3(1), dim3(1024), 0, s, status, B, idx_out, count, budget, next_chunk);
}

// Ready list of a pool-scheduled early-exit round: the candidates that still have a chunk to run (status OK, chunks_run < their
// budget), ordered by (chunks finished, candidate) -- the ones furthest behind first, so nobody starves at the end of the
// list -- with, per entry, the row of the (B * n_chunks, ...) noise view its next denoiser call reads.  One workgroup; one
// ordered prefix-sum pass per chunk index (n_chunks <= 64, B <= a few 10 000: microseconds).
__global__ void __launch_bounds__(1024) compact_ready_kernel(const int32_t* __restrict__ status, const int32_t* __restrict__ chunks_run,
                                                             const int32_t* __restrict__ budget, int n_chunks, int B,
                                                             int32_t* __restrict__ idx_out, int32_t* __restrict__ nrow_out,
                                                             int32_t* __restrict__ count) {
  __shared__ int s_wave_sum[16];
  const int tid = threadIdx.x;
  int base = 0;
  for (int j = 0; j < n_chunks; ++j) {
    for (int start = 0; start < B; start += blockDim.x) {
      const int b = start + tid;
      const int rdy = (b < B && status[b] == DITREE_ST_OK && chunks_run[b] == j && j < (budget ? budget[b] : n_chunks)) ? 1 : 0;
      int v = rdy;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d);
        if ((tid & 63) >= d) v += o;
      }
      if ((tid & 63) == 63) s_wave_sum[tid >> 6] = v;
      __syncthreads();
      int woff = 0, total = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        if (w < (tid >> 6)) woff += s_wave_sum[w];
        total += s_wave_sum[w];
      }
      if (rdy) {
        idx_out[base + woff + v - 1] = b;
        nrow_out[base + woff + v - 1] = b * n_chunks + j;
      }
      base += total;
      __syncthreads();
    }
  }
  if (tid == 0) *count = base;
}
void launch_compact_ready(const int32_t* status, const int32_t* chunks_run, const int32_t* budget, int n_chunks, int B,
                          int32_t* idx_out, int32_t* nrow_out, int32_t* count, hipStream_t s) {
  hipLaunchKernelGGL(compact_ready_kernel, dim3(1), dim3(1024), 0, s, status, chunks_run, budget, n_chunks, B, idx_out, nrow_out, count);
}

// planners/RRT.py:149-152 for a round: candidate b is visit number num_visit[parent] + (earlier candidates of the round with
// the same parent) of its parent; its edge runs schedule[clip(visit)] chunks.  B^2 / 2 parent compares out of L2 (B <= 8192).
struct ScheduleArg { int n; int chunks[16]; };
__global__ void __launch_bounds__(256) chunk_budget_kernel(const int32_t* __restrict__ parent, int B,
                                                           const int32_t* __restrict__ num_visit, ScheduleArg sc,
                                                           int32_t* __restrict__ budget) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int p = parent[b];
  int k = num_visit[p];
  for (int q = 0; q < b; ++q) k += (parent[q] == p) ? 1 : 0;
  k = k < 0 ? 0 : (k > sc.n - 1 ? sc.n - 1 : k);
  budget[b] = sc.chunks[k];
}
void launch_chunk_budget(const int32_t* parent, int B, const int32_t* num_visit, const int32_t* chunks, int n,
                         int32_t* budget, hipStream_t s) {
  ScheduleArg sc{};
  sc.n = n;
  for (int i = 0; i < n && i < 16; ++i) sc.chunks[i] = chunks[i];
  hipLaunchKernelGGL(chunk_budget_kernel, dim3((B + 255) / 256), dim3(256), 0, s, parent, B, num_visit, sc, budget);
}

// rows of a strided f32 matrix gathered by candidate index (the chunk's noise for the alive candidates)
__global__ void gather_rows_f32_kernel(const float* __restrict__ src, int64_t src_stride, const int32_t* __restrict__ idx,
                                       float* __restrict__ dst, int row_floats, int n) {
  const int r = blockIdx.x;
  if (r >= n) return;
  const float* s = src + (size_t)idx[r] * src_stride;
  float* d = dst + (size_t)r * row_floats;
  for (int k = threadIdx.x; k < row_floats; k += blockDim.x) d[k] = s[k];
}
void launch_gather_rows_f32(const float* src, int64_t src_stride, const int32_t* idx, float* dst, int row_floats, int n,
                            hipStream_t s) {
  hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(n), dim3(128), 0, s, src, src_stride, idx, dst, row_floats, n);
}

// ------------------------------------------------------------------------- obstacle ahead
// planners/RRT.py:61-81: samples = linspace(0, 1.5, 30)[:, None] @ [[cos(-psi), sin(-psi)]] + (col, row) with
// (row, col) = cell_xy_to_rowcol(xy, floor_enable=False); astype('int') truncates toward zero; clip; any.
__device__ __forceinline__ bool obstacle_ahead_dev(double x, double y, double psi, const unsigned char* mz, int H, int W,
                                                   const AheadArg& ts) {
  const double row = ((double)H / 2.0 - y) / 1.0, col = (x + (double)W / 2.0) / 1.0;     // car_env.py:196-201
  double c, sn;
  sincos(-psi, &sn, &c);
  bool any = false;
#pragma unroll 1
  for (int i = 0; i < 30; ++i) {
    const double px = ts.t[i] * c + col, py = ts.t[i] * sn + row;
    // numpy's astype(int) of NaN / out-of-range is INT64_MIN -> clipped to 0
    long long qx = (px == px && fabs(px) < 9.0e18) ? (long long)px : (long long)-1;
    long long qy = (py == py && fabs(py) < 9.0e18) ? (long long)py : (long long)-1;
    const int ix = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : (int)qx);
    const int iy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : (int)qy);
    any |= mz[iy * W + ix] != 0;
  }
  return any;
}
__global__ void obstacle_ahead_kernel(const unsigned char* __restrict__ maze, int rows, int cols,
                                      const double* __restrict__ state, int stride, int B, AheadArg ts,
                                      uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  stage_maze(lds, maze, rows * cols);
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* s = state + (size_t)b * stride;
  out[b] = obstacle_ahead_dev(s[0], s[1], s[2], lds, rows, cols, ts) ? 1 : 0;
}
void launch_obstacle_ahead(const unsigned char* maze, int rows, int cols, const double* state, int stride, int B,
                           const AheadArg& ts, uint8_t* out, hipStream_t s) {
  size_t lds = ((size_t)rows * cols + 15) & ~(size_t)15;
  hipLaunchKernelGGL(obstacle_ahead_kernel, dim3((B + 255) / 256), dim3(256), lds, s, maze, rows, cols, state, stride, B,
                     ts, out);
}

// ------------------------------------------------------------------------- fallback selection
// planners/RRT.py:233-254 over nodes 1..n-1.  One workgroup; every thread scans a strided subset and keeps
// (key, index) with first-occurrence ties, then a block reduction.
__global__ void __launch_bounds__(1024) fallback_select_kernel(ditree_tree t, int n, double gx, double gy,
                                                               const double* __restrict__ path, int P,
                                                               int32_t* __restrict__ out_node) {
  __shared__ double s_key[1024];
  __shared__ int s_idx[1024];
  const int tid = threadIdx.x;
  // minimise `key`: cost for the goal-distance rule, -progress for the along-path rule
  double best = __builtin_huge_val();
  int bidx = 0x7fffffff;
  int free_seen = 0;
  for (int i = 1 + tid; i < n; i += blockDim.x) {
    const double x = t.xy[(size_t)i * 2], y = t.xy[(size_t)i * 2 + 1];
    const bool obs = t.obstacle_ahead != nullptr && t.obstacle_ahead[i] != 0;
    free_seen |= !obs;
    double key;
    if (path == nullptr) {
      const double dx = x - gx, dy = y - gy;
      key = sqrt(fma(dy, dy, dx * dx)) + 10e3 * (obs ? 1.0 : 0.0);      // np.linalg.norm (1-D) + 10e3 * flag
    } else {
      int prog = -1;
      if (!obs) {
        double bd = __builtin_huge_val();
        for (int k = 0; k < P; ++k) {
          const double dx = x - path[2 * k], dy = y - path[2 * k + 1];
          const double d = sqrt(dx * dx + dy * dy);                      // np.linalg.norm(axis=1): no fused dot
          if (d < bd) { bd = d; prog = k; }
        }
      }
      key = -(double)prog;
    }
    if (key < best) { best = key; bidx = i; }
  }
  s_key[tid] = best;
  s_idx[tid] = bidx;
  const int any_free = __syncthreads_or(free_seen);
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if (tid < off) {
      const double ok = s_key[tid + off];
      const int oi = s_idx[tid + off];
      if (ok < s_key[tid] || (ok == s_key[tid] && oi < s_idx[tid])) { s_key[tid] = ok; s_idx[tid] = oi; }
    }
    __syncthreads();
  }
  // RRT.py:227-232: np.all(has_obstacle_ahead) (true for an empty list) -> no plan
  if (tid == 0) *out_node = (!any_free || s_idx[0] == 0x7fffffff) ? -1 : s_idx[0];
}
void launch_fallback_select(const ditree_tree& t, int n_nodes, double gx, double gy, const double* path_dev, int P,
                            int32_t* out_node, hipStream_t s) {
  hipLaunchKernelGGL(fallback_select_kernel, dim3(1), dim3(1024), 0, s, t, n_nodes, gx, gy, path_dev, P, out_node);
}

// ------------------------------------------------------------------------- accept + commit
// Phase 1 (one workgroup): the reference's sequential accept order (RRT.py:179-217) as a
// scan.  counters: [0] n_nodes [1] goal node [2] env.done latched [3] chunk iterations
// [4] candidates [5] sticky triggered [6] capacity overflow [7] phantom candidate (scratch).
__global__ void __launch_bounds__(1024)
accept_scan_kernel(ditree_tree t, ditree_round r, int emulate_sticky) {
  __shared__ int s_first_goal, s_first_gac, s_total, s_iters;
  __shared__ int s_wave_sum[16];
  const int B = r.B;
  const int tid = threadIdx.x;
  if (tid == 0) { s_first_goal = 0x7fffffff; s_first_gac = 0x7fffffff; s_total = 0; s_iters = 0; }
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) {
    int st = r.status[b];
    if ((st & 0xff) == DITREE_ST_GOAL) atomicMin(&s_first_goal, b);
    if (emulate_sticky && (st & 0xff) == DITREE_ST_COLLIDED && (st & DITREE_ST_FLAG_GOAL_AT_COLLISION))
      atomicMin(&s_first_gac, b);
  }
  __syncthreads();
  const int g = s_first_goal, c = s_first_gac;
  const bool latched_in = emulate_sticky && t.counters[2] != 0;
  int last = B - 1, phantom = -1, goal_cand = -1;
  bool latch_out = latched_in;
  if (latched_in) {
    phantom = 0; last = 0; goal_cand = 0;
  } else if (c != 0x7fffffff && (g == 0x7fffffff || g > c)) {
    if (c + 1 < B) { phantom = c + 1; last = c + 1; goal_cand = c + 1; latch_out = true; }
    else { latch_out = true; }
  } else if (g != 0x7fffffff) {
    last = g; goal_cand = g;
  }
  const int n0 = t.counters[0];
  // ordered prefix sum of accepted flags over candidates 0..last, 1024 at a time
  int base = 0;
  for (int start = 0; start <= last; start += blockDim.x) {
    int b = start + tid;
    int acc = 0, it = 0;
    if (b <= last) {
      int st = r.status[b] & 0xff;
      acc = (b == phantom) ? 1 : (st != DITREE_ST_COLLIDED);
      it = (b == phantom) ? 1 : r.chunks_run[b];
      atomicAdd(&t.num_visit[r.parent[b]], 1);
    }
    // wave inclusive scan
    int v = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int o = __shfl_up(v, d);
      if ((tid & 63) >= d) v += o;
    }
    int itw = it;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) itw += __shfl_xor(itw, m);
    if ((tid & 63) == 63) s_wave_sum[tid >> 6] = v;
    if ((tid & 63) == 0) atomicAdd(&s_iters, itw);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (tid >> 6); ++w) woff += s_wave_sum[w];
    int chunk_total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) chunk_total += s_wave_sum[w];
    if (b <= last) {
      int rank = base + woff + v - acc;                 // exclusive rank
      int id = acc ? (n0 + rank) : -1;
      if (id >= t.capacity) id = -1;
      r.node_id[b] = id;
    }
    base += chunk_total;
    __syncthreads();
  }
  for (int b = last + 1 + tid; b < B; b += blockDim.x) r.node_id[b] = -1;
  __syncthreads();
  if (tid == 0) {
    int total = base;
    int n1 = n0 + total;
    if (n1 > t.capacity) { n1 = t.capacity; t.counters[6] = 1; }
    t.counters[0] = n1;
    if (goal_cand >= 0) {
      int gid = r.node_id[goal_cand];
      t.counters[1] = gid;
    }
    t.counters[2] = (emulate_sticky && latch_out && phantom < 0) ? 1 : ((phantom >= 0) ? 1 : t.counters[2]);
    t.counters[3] += s_iters;
    t.counters[4] += last + 1;
    if (phantom >= 0) t.counters[5] = 1;
    t.counters[7] = phantom;
  }
}

// Phase 2: one wave per candidate copies the accepted edge into its node slot, dropping
// all-zero rows (RRT.py:196-199).  State / action width S / D at run time (car 6 / 2, ant 29 / 8; S, D <= 64: a lane per
// component, the zero-row test is one ballot).
__global__ void __launch_bounds__(64)
accept_commit_kernel(ditree_tree t, ditree_round r, const unsigned char* __restrict__ maze, int rows, int cols,
                     AheadArg ts) {
  const int b = blockIdx.x;
  const int id = r.node_id[b];
  if (id < 0) return;
  const int lane = threadIdx.x;
  const int A = t.A, nC = t.n_chunks, S = t.state_dim, D = t.action_dim;
  const int phantom = t.counters[7];
  const int par = r.parent[b];
  const size_t es_cap = (size_t)nC * (A + 1), ea_cap = (size_t)nC * A;
  double* es = t.edge_states + (size_t)id * es_cap * S;
  double* ea = t.edge_actions + (size_t)id * ea_cap * D;
  const double* cs = r.states + (size_t)b * es_cap * S;
  const double* ca = r.actions + (size_t)b * ea_cap * D;
  int ns = 0, na = 0;
  double endv = 0.0;                                   // lane k < S: component k of the node's state
  // sharded rounds: the trajectories of candidates another rank expanded are not here -- their nodes get state, parent
  // and last action from the exchanged record, the edge rows stay with the owner (edge_owner)
  const bool own = r.shard == 0 || (b >= r.own_lo && b < r.own_lo + r.own_n);
  int owner = -1;
  if (b == phantom) {
    // frozen env step (car_env.py:254): the edge is [s, s] and the first sampled action
    if (lane < S) endv = t.state[(size_t)par * S + lane];
    const bool zero = __ballot(lane < S && endv != 0.0) == 0ull;
    if (!zero) {
      if (lane < S) { es[lane] = endv; es[S + lane] = endv; }
      ns = 2;
    }
    double fa = 0.0;
    if (lane < D) fa = r.first_action ? r.first_action[(size_t)b * D + lane] : ca[lane];
    if (__ballot(lane < D && fa != 0.0) != 0ull) {
      if (lane < D) ea[lane] = fa;
      na = 1;
    }
  } else if (!own) {
    if (lane < S) endv = r.end_state[(size_t)b * S + lane];
    owner = r.shard > 0 ? b / r.shard : 0;
    ns = -1;
    na = -1;
  } else {
    if (lane < S) endv = r.end_state[(size_t)b * S + lane];
    owner = r.shard > 0 ? b / r.shard : -1;
    const int run = r.chunks_run[b];
    const int rows_s = run * (A + 1), rows_a = run * A;
    // serial compaction per wave: rows are few (<= 72) and small
    for (int row = 0; row < rows_s; ++row) {
      const double v = lane < S ? cs[(size_t)row * S + lane] : 0.0;
      if (__ballot(v != 0.0) != 0ull) {
        if (lane < S) es[(size_t)ns * S + lane] = v;
        ++ns;
      }
    }
    for (int row = 0; row < rows_a; ++row) {
      const double v = lane < D ? ca[(size_t)row * D + lane] : 0.0;
      if (__ballot(v != 0.0) != 0ull) {
        if (lane < D) ea[(size_t)na * D + lane] = v;
        ++na;
      }
    }
  }
  if (lane < S) t.state[(size_t)id * S + lane] = endv;
  if (lane < 2) t.xy[(size_t)id * 2 + lane] = endv;
  const double psi = __shfl(endv, 2), ex = __shfl(endv, 0), ey = __shfl(endv, 1);
  if (lane == 0) {
    t.parent[id] = par;
    t.has_prev[id] = 1;
    t.num_visit[id] = 0;
    t.edge_nstates[id] = ns;
    t.edge_nactions[id] = na;
    if (t.edge_owner != nullptr) t.edge_owner[id] = owner;
    if (t.obstacle_ahead != nullptr)                                   // RRT.py:202-205 (run_type > 0)
      t.obstacle_ahead[id] = obstacle_ahead_dev(ex, ey, psi, maze, rows, cols, ts) ? 1 : 0;
  }
  if (lane < D) {
    double la;
    // a lane reads back what it stored itself (same address, program order)
    if (r.last_action != nullptr && b != phantom) la = r.last_action[(size_t)b * D + lane];     // exchanged record
    else la = (na > 0) ? ea[(size_t)(na - 1) * D + lane] : 0.0;
    t.last_action[(size_t)id * D + lane] = la;
  }
  if (t.hist != nullptr) {
    // what the first sampler call of a child will see (RRT.py:146-147: parent_states_seq, fm_policy.py:96-102: its last
    // obs_history = 3 rows): valid rows at the END of the node's three slots
    double* h = t.hist + (size_t)id * 3 * S;
    int n;
    if (ns >= 0) {
      n = ns < 3 ? ns : 3;
      for (int q = 0; q < n; ++q)
        if (lane < S) h[(size_t)(3 - n + q) * S + lane] = es[(size_t)(ns - n + q) * S + lane];
    } else {
      n = r.hist_n[b];
      for (int q = 3 - n; q < 3; ++q)
        if (lane < S) h[(size_t)q * S + lane] = r.hist[((size_t)b * 3 + q) * S + lane];
    }
    if (lane == 0) t.hist_n[id] = n;
  }
}

// ---- candidate records of a sharded round (include/ditree.h "Candidate record"): R = S + 2 D + 2 [+ 3 S] doubles
__device__ __forceinline__ bool row_is_zero(const double* p, int n) {
  bool z = true;
  for (int k = 0; k < n; ++k) z &= (p[k] == 0.0);
  return z;
}
__global__ void round_pack_kernel(ditree_tree t, ditree_round r, double* __restrict__ rec, int R) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= r.B) return;
  const int A = t.A, nC = t.n_chunks, S = t.state_dim, D = t.action_dim;
  double* o = rec + (size_t)b * R;
  for (int k = 0; k < S; ++k) o[k] = r.end_state[(size_t)b * S + k];
  const double* ca = r.actions + (size_t)b * nC * A * D;
  // last kept (non-zero) action row of the chunks that ran: what accept_commit_kernel derives from the compacted edge
  double* la = o + S;
  for (int k = 0; k < D; ++k) la[k] = 0.0;
  for (int row = r.chunks_run[b] * A - 1; row >= 0; --row) {
    if (!row_is_zero(ca + (size_t)row * D, D)) {
      for (int k = 0; k < D; ++k) la[k] = ca[(size_t)row * D + k];
      break;
    }
  }
  for (int k = 0; k < D; ++k) o[S + D + k] = ca[k];
  int n = 0;
  if (t.hist != nullptr) {
    // the last <= 3 kept state rows, valid rows at the end of the three slots
    const double* cs = r.states + (size_t)b * nC * (A + 1) * S;
    double* h = o + S + 2 * D + 2;
    for (int k = 0; k < 3 * S; ++k) h[k] = 0.0;
    for (int row = r.chunks_run[b] * (A + 1) - 1; row >= 0 && n < 3; --row) {
      if (!row_is_zero(cs + (size_t)row * S, S)) {
        for (int k = 0; k < S; ++k) h[(size_t)(2 - n) * S + k] = cs[(size_t)row * S + k];
        ++n;
      }
    }
  }
  int32_t* oi = (int32_t*)(o + S + 2 * D);
  oi[0] = r.parent[b]; oi[1] = r.status[b]; oi[2] = r.chunks_run[b]; oi[3] = n;
}
__global__ void round_unpack_kernel(ditree_tree t, ditree_round r, const double* __restrict__ rec, int R) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= r.B) return;
  const int S = t.state_dim, D = t.action_dim;
  const double* o = rec + (size_t)b * R;
  for (int k = 0; k < S; ++k) r.end_state[(size_t)b * S + k] = o[k];
  for (int k = 0; k < D; ++k) {
    r.last_action[(size_t)b * D + k] = o[S + k];
    r.first_action[(size_t)b * D + k] = o[S + D + k];
  }
  const int32_t* oi = (const int32_t*)(o + S + 2 * D);
  r.parent[b] = oi[0]; r.status[b] = oi[1]; r.chunks_run[b] = oi[2];
  if (t.hist != nullptr) {
    r.hist_n[b] = oi[3];
    const double* h = o + S + 2 * D + 2;
    for (int k = 0; k < 3 * S; ++k) r.hist[(size_t)b * 3 * S + k] = h[k];
  }
}
int record_doubles(const ditree_tree& t) { return t.state_dim + 2 * t.action_dim + 2 + (t.hist ? 3 * t.state_dim : 0); }
void launch_round_pack(const ditree_tree& t, const ditree_round& r, double* rec, hipStream_t s) {
  hipLaunchKernelGGL(round_pack_kernel, dim3((r.B + 127) / 128), dim3(128), 0, s, t, r, rec, record_doubles(t));
}
void launch_round_unpack(const ditree_tree& t, const ditree_round& r, const double* rec, hipStream_t s) {
  hipLaunchKernelGGL(round_unpack_kernel, dim3((r.B + 127) / 128), dim3(128), 0, s, t, r, rec, record_doubles(t));
}

void launch_accept(const ditree_tree& t, const ditree_round& r, int emulate_sticky, const unsigned char* maze, int rows,
                   int cols, const AheadArg& ts, hipStream_t s) {
  hipLaunchKernelGGL(accept_scan_kernel, dim3(1), dim3(1024), 0, s, t, r, emulate_sticky);
  hipLaunchKernelGGL(accept_commit_kernel, dim3(r.B), dim3(64), 0, s, t, r, maze, rows, cols, ts);
}
