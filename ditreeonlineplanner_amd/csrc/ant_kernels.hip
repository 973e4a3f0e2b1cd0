// Ant (BASELINE config 3) kernels of the expansion path: collision glue, the higher-DoF rollout slot, history gather.
// Compiled with -ffp-contract=off (flags bit-exact against the CPU oracle).
//
// Reference sites (paths relative to the reference root):
//   ant_collision   common/map_utils.py:126-219 as called at planners/base_planner.py:154-155
//   ant_rollout     planners/base_planner.py:257-320 (ant branches :278-279,296-298) -- the env step itself is MuJoCo
//                   (third party, no oracle): a next-observation tape or the build's stand-in model takes its place
//   ant_gather      planners/RRT.py:144-147 (prev_actions / prev_states of the first sampler call of an edge)
#include "ant_device.h"
#include "ditree_internal.h"

__device__ __forceinline__ void stage_maze_ant(unsigned char* lds, const unsigned char* __restrict__ g, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
}

// ------------------------------------------------------------------------- collision
__global__ void __launch_bounds__(256)
ant_collision_kernel(const unsigned char* __restrict__ maze, int rows, int cols, const double* __restrict__ state, int stride, int B,
                     double ball_radius, double s_global, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  stage_maze_ant(lds, maze, rows * cols);
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  out[b] = ant_collides(state + (size_t)b * stride, lds, rows, cols, s_global, ball_radius) ? 1 : 0;
}
void launch_ant_collision(const unsigned char* maze, int rows, int cols, const double* state, int stride, int B, double ball_radius,
                          double s_global, uint8_t* out, hipStream_t s) {
  const size_t lds = ((size_t)rows * cols + 15) & ~(size_t)15;
  hipLaunchKernelGGL(ant_collision_kernel, dim3((B + 255) / 256), dim3(256), lds, s, maze, rows, cols, state, stride, B, ball_radius,
                     s_global, out);
}

// ------------------------------------------------------------------------- history gather (round begin)
// hist (B, 3, 29) <- tree.hist[parent], hist_n <- tree.hist_n[parent]; one thread per (candidate, element).
__global__ void ant_gather_hist_kernel(const int32_t* __restrict__ parent, const double* __restrict__ node_hist,
                                       const int32_t* __restrict__ node_hist_n, int B, double* __restrict__ hist,
                                       int32_t* __restrict__ hist_n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = e / (3 * ANT_S), k = e - b * (3 * ANT_S);
  if (b >= B) return;
  const int p = parent[b];
  hist[(size_t)b * 3 * ANT_S + k] = node_hist[(size_t)p * 3 * ANT_S + k];
  if (k == 0) hist_n[b] = node_hist_n[p];
}
void launch_ant_gather_hist(const int32_t* parent, const double* node_hist, const int32_t* node_hist_n, int B, double* hist,
                            int32_t* hist_n, hipStream_t s) {
  const int n = B * 3 * ANT_S;
  hipLaunchKernelGGL(ant_gather_hist_kernel, dim3((n + 255) / 256), dim3(256), 0, s, parent, node_hist, node_hist_n, B, hist, hist_n);
}

// round->actions[b, j, :A, :] <- the first A rows of candidate b's sampled sequence (what a host-side simulator reads before it
// steps; the rollout kernel rewrites the same rows, zeroing those behind a goal step).
__global__ void ant_copy_actions_kernel(const double* __restrict__ act, int64_t act_stride, int act_dense,
                                        const int32_t* __restrict__ idx, const int32_t* __restrict__ status, int n_run, int A,
                                        double* __restrict__ out, int64_t out_stride) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int ob = e / (A * ANT_D), k = e - ob * (A * ANT_D);
  if (ob >= n_run) return;
  const int b = idx ? idx[ob] : ob;
  if (status[b] != DITREE_ST_OK) return;
  out[(size_t)b * out_stride + k] = act[(size_t)(act_dense ? ob : b) * act_stride + k];
}
void launch_ant_copy_actions(const double* act, int64_t act_stride, int act_dense, const int32_t* idx, const int32_t* status,
                             int n_run, int A, double* out, int64_t out_stride, hipStream_t s) {
  const int n = n_run * A * ANT_D;
  hipLaunchKernelGGL(ant_copy_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, s, act, act_stride, act_dense, idx, status, n_run,
                     A, out, out_stride);
}

// ------------------------------------------------------------------------- rollout (the higher-DoF dynamics slot)
// One lane per candidate: 29 doubles of state in registers, A env steps, after each the goal + collision test.  MODEL: the env
// step is ant_model_step (frame_skip sub-steps of the stand-in model); else row i of the candidate's next-observation tape.
// Row outputs are addressed through ditree_strides: packed per candidate inside a round (the accept kernels read candidate
// rows), step-major / component-major / candidate-minor for the standalone rollouts (a wave's 64 stores of one component are 512
// contiguous bytes; 29-double AoS rows would touch 64 sectors per store).
//   hist_out (B, 3, 29) / hist_n: `prev_states = curr_states_seq` (RRT.py:190) -- the last min(3, A + 1) rows of [start, obs_1
//   .. obs_A] land at the END of the candidate's three slots (only meaningful when the chunk ends with status OK).
template <bool MODEL, bool STAGE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))   // the action burst holds 64 registers
ant_rollout_kernel(const unsigned char* __restrict__ maze, int rows, int cols, AntModelArg m, double* __restrict__ state_io,
                   const double* __restrict__ actions, int64_t act_stride, const double* __restrict__ tape, int64_t tape_stride,
                   int32_t* __restrict__ status_io, int B, int A, double gx, double gy, double goal_radius, double ball_radius,
                   double s_global, double* __restrict__ states_out, ditree_strides sl, double* __restrict__ actions_out,
                   ditree_strides al, int32_t* __restrict__ steps_out, int64_t steps_stride, int32_t* __restrict__ chunks_run,
                   double* __restrict__ prev_action_io, uint8_t* __restrict__ has_prev_io, double* __restrict__ hist_out,
                   int32_t* __restrict__ hist_n, const int32_t* __restrict__ idx, int act_dense, int chunk_from_counter,
                   AntChunkStrides cs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  stage_maze_ant(lds, maze, rows * cols);
  const int ob = blockIdx.x * blockDim.x + threadIdx.x;
  if (ob >= B) return;
  const int b = idx ? idx[ob] : ob;
  if (status_io[b] != DITREE_ST_OK) return;
  if (chunk_from_counter) {                               // pool-scheduled early-exit rounds: rows of one launch at different chunks
    const int j = chunks_run[b];
    if (states_out) states_out += (size_t)j * cs.states;
    if (actions_out) actions_out += (size_t)j * cs.actions_out;
    if (steps_out) steps_out += j;
    if (!act_dense) actions += (size_t)j * cs.actions_in;
    if (!MODEL) tape += (size_t)j * cs.tape;
  }
  double s[ANT_S];
#pragma unroll
  for (int k = 0; k < ANT_S; ++k) s[k] = state_io[(size_t)b * ANT_S + k];
  const double* act = actions + (size_t)(act_dense ? ob : b) * act_stride;
  const double* tp = MODEL ? nullptr : tape + (size_t)b * tape_stride;
  double* so = states_out ? states_out + (size_t)b * sl.cand : nullptr;
  double* ao = actions_out ? actions_out + (size_t)b * al.cand : nullptr;
  double* ho = hist_out ? hist_out + (size_t)b * 3 * ANT_S : nullptr;
  const int first_kept = A + 1 - 3;                       // row r of the chunk goes to history slot r - first_kept (if >= 0)
  if (so) {
#pragma unroll
    for (int k = 0; k < ANT_S; ++k) so[k * sl.comp] = s[k];
  }
  if (ho && 0 - first_kept >= 0) {
#pragma unroll
    for (int k = 0; k < ANT_S; ++k) ho[(size_t)(0 - first_kept) * ANT_S + k] = s[k];
  }
  int status = DITREE_ST_OK, steps = 0;
  double a[ANT_D], la[ANT_D];
#pragma unroll
  for (int k = 0; k < ANT_D; ++k) la[k] = 0.0;
  // STAGE: the action rows of four steps (16 x 16 bytes) come in one burst of loads into the lane's LDS slots (slot q of lane t at
  // (q * blockDim + t) * 16) and are read back step by step; and every load issued so far has arrived before the loop starts.
  // vmcnt retires in order and counts stores: a per-step action load -- or a late wait for the state loaded above -- made every
  // step wait for the row stores of its predecessor (car_rollout_kernel, geom_kernels.hip, has the measurements).
  double2* abuf = STAGE ? reinterpret_cast<double2*>(lds + (((size_t)rows * cols + 15) & ~(size_t)15)) + threadIdx.x : nullptr;
  const int bd = blockDim.x;
  __builtin_amdgcn_s_waitcnt(0x0F70);                                            // vmcnt(0)
  // Rows are stored in LOCKSTEP (every lane stores row i + 1 in the same instruction: its state while the edge runs, the zero
  // row of base_planner.py:282 afterwards), so candidate-minor storage writes whole 512-byte runs (see car_rollout_kernel).
  bool alive = true;
  for (int i = 0; i < A; ++i) {
    if constexpr (STAGE) {
      if ((i & 3) == 0) {
        // sixteen named values, not an array: the scheduling barrier is opaque to the optimiser, an array on its two sides would
        // stay in scratch memory
#define ACT_LD(q) const double2 t##q = *reinterpret_cast<const double2*>(act + (size_t)min(i + (q >> 2), A - 1) * ANT_D + 2 * (q & 3))
        ACT_LD(0); ACT_LD(1); ACT_LD(2); ACT_LD(3); ACT_LD(4); ACT_LD(5); ACT_LD(6); ACT_LD(7);
        ACT_LD(8); ACT_LD(9); ACT_LD(10); ACT_LD(11); ACT_LD(12); ACT_LD(13); ACT_LD(14); ACT_LD(15);
#undef ACT_LD
        __builtin_amdgcn_sched_barrier(0);                        // all sixteen loads in flight before the first LDS write waits
#define ACT_ST(q) abuf[q * bd] = t##q
        ACT_ST(0); ACT_ST(1); ACT_ST(2); ACT_ST(3); ACT_ST(4); ACT_ST(5); ACT_ST(6); ACT_ST(7);
        ACT_ST(8); ACT_ST(9); ACT_ST(10); ACT_ST(11); ACT_ST(12); ACT_ST(13); ACT_ST(14); ACT_ST(15);
#undef ACT_ST
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 v = abuf[((i & 3) * 4 + k) * bd];
        a[2 * k] = v.x; a[2 * k + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < ANT_D; ++k) a[k] = act[(size_t)i * ANT_D + k];
    }
    if (alive) {
      if constexpr (MODEL) {
        ant_model_step(s, a, m);
      } else {
#pragma unroll
        for (int k = 0; k < ANT_S; ++k) s[k] = tp[(size_t)i * ANT_S + k];
      }
      steps = i + 1;
#pragma unroll
      for (int k = 0; k < ANT_D; ++k) la[k] = a[k];
    } else if (status == DITREE_ST_GOAL) {
#pragma unroll
      for (int k = 0; k < ANT_D; ++k) a[k] = 0.0;          // remaining actions zeroed on the goal branch only (:315)
    }
    if (so) {
#pragma unroll
      for (int k = 0; k < ANT_S; ++k) so[(size_t)(i + 1) * sl.row + k * sl.comp] = alive ? s[k] : 0.0;
    }
    if (ho && alive && i + 1 - first_kept >= 0) {
#pragma unroll
      for (int k = 0; k < ANT_S; ++k) ho[(size_t)(i + 1 - first_kept) * ANT_S + k] = s[k];
    }
    if (ao) {
#pragma unroll
      for (int k = 0; k < ANT_D; ++k) ao[(size_t)i * al.row + k * al.comp] = a[k];
    }
    if (alive) {
      const double ex = s[0] - gx, ey = s[1] - gy;
      const bool done = sqrt(fma(ey, ey, ex * ex)) < goal_radius;      // np.linalg.norm (ddot = one fma), base_planner.py:296-297
      const bool coll = ant_collides(s, lds, rows, cols, s_global, ball_radius);   // :306 before :314
      if (coll) { status = DITREE_ST_COLLIDED; alive = false; }
      else if (done) { status = DITREE_ST_GOAL; alive = false; }
    }
    if (states_out == nullptr && actions_out == nullptr && !__any(alive)) break;
  }
#pragma unroll
  for (int k = 0; k < ANT_S; ++k) state_io[(size_t)b * ANT_S + k] = s[k];
  status_io[b] = status;
  if (steps_out) steps_out[(size_t)b * steps_stride] = steps;
  if (chunks_run) chunks_run[b] += 1;
  if (status == DITREE_ST_OK) {
    if (prev_action_io) {                                 // RRT.py:188 prev_actions = curr_action_seq (its last row conditions)
#pragma unroll
      for (int k = 0; k < ANT_D; ++k) prev_action_io[(size_t)b * ANT_D + k] = la[k];
      if (has_prev_io) has_prev_io[b] = 1;
    }
    if (hist_n) hist_n[b] = A + 1 < 3 ? A + 1 : 3;
  }
}

void launch_ant_rollout(const unsigned char* maze, int rows, int cols, const AntModelArg* model, double* state_io,
                        const double* actions, int64_t act_stride, const double* tape, int64_t tape_stride, int32_t* status_io, int B,
                        int A, double gx, double gy, double goal_radius, double ball_radius, double s_global, double* states_out,
                        ditree_strides sl, double* actions_out, ditree_strides al, int32_t* steps_out, int64_t steps_stride,
                        int32_t* chunks_run, double* prev_action_io, uint8_t* has_prev_io, double* hist_out, int32_t* hist_n,
                        const int32_t* idx, int act_dense, hipStream_t s, int chunk_from_counter, AntChunkStrides cs) {
  const size_t lds = ((size_t)rows * cols + 15) & ~(size_t)15;
  // one wave per work-group while the batch is small (a round's 4096 candidates: 64 CUs instead of 16), four once every SIMD has
  // a wave anyway (65 536 x 16: 271.5 us against 279.2 in groups of 64).  Two-wave groups are the one size to avoid
  // (car_rollout_kernel: +27 %); DITREE_ROLLOUT_BLK = 64 | 256 overrides.
  static int blk_env = -1;
  if (blk_env < 0) { const char* e = getenv("DITREE_ROLLOUT_BLK"); blk_env = e ? atoi(e) : 0; }
  const int blk = (blk_env == 64 || blk_env == 256) ? blk_env : (B >= 32768 ? 256 : 64);
  static bool attr_done_dev[64] = {};                        // the attribute is per kernel AND per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool attr_done = dev >= 0 && dev < 64 && attr_done_dev[dev];
  if (!attr_done) {                                          // 256 threads x 256 B of staged actions + the maze exceed 64 KB
    const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
    (void)hipFuncSetAttribute((const void*)ant_rollout_kernel<true, true>, at, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ant_rollout_kernel<false, true>, at, 160 * 1024);
    if (dev >= 0 && dev < 64) attr_done_dev[dev] = true;
  }
  const dim3 grid((B + blk - 1) / blk);
  static int stage_env = -1;
  if (stage_env < 0) { const char* e = getenv("DITREE_ROLLOUT_STAGE"); stage_env = e ? atoi(e) : 1; }
  bool stage = stage_env && (act_stride % 2 == 0) && (cs.actions_in % 2 == 0) && ((uintptr_t)actions % 16 == 0);
  if (lds + (size_t)blk * 256 > 160 * 1024) stage = false;    // a maze that leaves no room for the staged actions: per-step loads
#define ANT_ROLLOUT_LAUNCH(MM, SS, MODEL_ARG)                                                                                     \
  hipLaunchKernelGGL((ant_rollout_kernel<MM, SS>), grid, dim3(blk), lds + (SS ? (size_t)blk * 256 : 0), s, maze, rows, cols, MODEL_ARG, \
                     state_io, actions, act_stride, tape, tape_stride, status_io, B, A, gx, gy, goal_radius, ball_radius, s_global,  \
                     states_out, sl, actions_out, al, steps_out, steps_stride, chunks_run, prev_action_io, has_prev_io, hist_out,    \
                     hist_n, idx, act_dense, chunk_from_counter, cs)
  if (model) { if (stage) ANT_ROLLOUT_LAUNCH(true, true, *model); else ANT_ROLLOUT_LAUNCH(true, false, *model); }
  else { if (stage) ANT_ROLLOUT_LAUNCH(false, true, AntModelArg{}); else ANT_ROLLOUT_LAUNCH(false, false, AntModelArg{}); }
#undef ANT_ROLLOUT_LAUNCH
}
