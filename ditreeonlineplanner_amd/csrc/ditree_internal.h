// Internal declarations shared by the HIP translation units of libditree_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ditree.h"

#define DITREE_MAX_AXIS 64
#define DITREE_LIDAR_RAYS 181

struct AxisArg {
  double v[DITREE_MAX_AXIS];
};
struct NormArg {
  double obs_mean[6], obs_std[6], act_mean[2], act_std[2];
};

struct AntNormArg {
  double obs_mean[27], obs_std[27], act_mean[8], act_std[8];
};

struct DenoiserState;   // denoise_host.hip

struct ditree_ctx {
  int device = 0;
  std::string err;
  // known maze (u8 cell codes) on the device
  unsigned char* maze = nullptr;
  int rows = 0, cols = 0;
  size_t maze_cap = 0;
  // expand-round scratch (sized for scratch_B candidates)
  int scratch_B = 0;
  double* cur_state = nullptr;      // (B,6)
  double* prev_action = nullptr;    // (B,2)
  uint8_t* has_prev = nullptr;      // (B,)
  float* lmap = nullptr;            // (B,n,n)
  float* cond = nullptr;            // (B,7)
  double* act64 = nullptr;          // (B,P,2)
  int scratch_lm = 0, scratch_P = 0;
  int32_t* alive_idx = nullptr;     // (B,) compacted candidate indices (early-exit rounds)
  int32_t* alive_nrow = nullptr;    // (B,) their rows of the (B * n_chunks, P, D) noise view
  int32_t* alive_cnt = nullptr;     // device scalar
  int ee_calls = 0, ee_waves = 0;   // denoiser calls / tile-waves of the last early-exit round (ditree_round_stats)
  int32_t* alive_cnt_host = nullptr;  // pinned host copy
  double* path_dev = nullptr;       // reference path xy for the fallback selection
  int path_cap = 0;
  // ant round scratch (ditree_expand_round_ant): (B, 3, 29) history + valid rows, (B, 8) previous action, flags, (B, 97) cond
  int ant_B = 0;
  double* ant_hist = nullptr;
  int32_t* ant_hist_n = nullptr;
  int32_t* ant_idx = nullptr;       // alive candidates of an early-exit round
  int32_t* ant_nrow = nullptr;      // their rows of the (B * n_chunks, P, 8) noise view
  int ant_n_run = 0;                // rows the current chunk runs on (ditree_ant_chunk_sample -> _step)
  const int32_t* ant_run_idx = nullptr;
  double* ant_prev = nullptr;
  uint8_t* ant_hasprev = nullptr;
  float* ant_cond = nullptr;
  float* ant_lmap = nullptr;
  double* ant_act = nullptr;
  int ant_P = 0, ant_lm = 0;
  double* mppi_partial = nullptr;   // partial sums of the MPPI update (ditree_mppi_step)
  double* mppi_ant_partial = nullptr;   // the same for ditree_mppi_step_ant (3 + 8 T per slice)
  unsigned long long* mppi_minkey = nullptr;   // running minimum of the rollout costs (order-preserving integer image)
  DenoiserState* dn = nullptr;
  // optional RCCL communicator (ditree_comm_*): librccl opened at run time
  void* rccl_lib = nullptr;
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
};

int set_err(ditree_ctx* ctx, int code, const std::string& msg);
#define HIP_TRY(ctx, expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return set_err(ctx, DITREE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// geom_kernels.hip launchers (all asynchronous on `s`)
void launch_maze_convert(const float* src, unsigned char* dst, int n, hipStream_t s);
void launch_nn_argmin(const double* queries, int q_stride, int B, const double* node_xy, int N,
                      int32_t* out_idx, const double* node_state, const double* node_last_action,
                      const uint8_t* node_has_prev, double* out_state, double* out_prev_action,
                      uint8_t* out_has_prev, hipStream_t s, int S = 6, int D = 2);
void launch_local_map(const unsigned char* maze, int rows, int cols, const double* state,
                      const int32_t* active, const int32_t* idx, int B, int n, const AxisArg& axis,
                      double s_global, int scaled, float* out, hipStream_t s, int state_stride = 6);
void launch_cond_vector_ant(const double* obs, int n_rows, const int32_t* hist_n, const double* prev_action, const uint8_t* has_prev,
                            const double* cond_goal, const int32_t* idx, int B, const AntNormArg& nm, double lm_size, float* out,
                            hipStream_t s);
void launch_cond_vector(const double* state, const double* prev_action, const uint8_t* has_prev,
                        const double* cond_goal, const int32_t* idx, int B, const NormArg& nm, double lm_size,
                        float* out, hipStream_t s);
void launch_path_after_obstacle(const float* path, int stride, int P, double cx, double cy, int f32_state, const unsigned char* maze,
                                int rows, int cols, int32_t* out, hipStream_t s);
// ant_kernels.hip
struct AntModelArg;
struct AntChunkStrides { int64_t states, actions_out, actions_in, tape; };
void launch_ant_collision(const unsigned char* maze, int rows, int cols, const double* state, int stride, int B, double ball_radius,
                          double s_global, uint8_t* out, hipStream_t s);
void launch_ant_gather_hist(const int32_t* parent, const double* node_hist, const int32_t* node_hist_n, int B, double* hist,
                            int32_t* hist_n, hipStream_t s);
void launch_ant_copy_actions(const double* act, int64_t act_stride, int act_dense, const int32_t* idx, const int32_t* status,
                             int n_run, int A, double* out, int64_t out_stride, hipStream_t s);
void launch_ant_rollout(const unsigned char* maze, int rows, int cols, const AntModelArg* model, double* state_io,
                        const double* actions, int64_t act_stride, const double* tape, int64_t tape_stride, int32_t* status_io, int B,
                        int A, double gx, double gy, double goal_radius, double ball_radius, double s_global, double* states_out,
                        ditree_strides sl, double* actions_out, ditree_strides al, int32_t* steps_out, int64_t steps_stride,
                        int32_t* chunks_run, double* prev_action_io, uint8_t* has_prev_io, double* hist_out, int32_t* hist_n,
                        const int32_t* idx, int act_dense, hipStream_t s, int chunk_from_counter = 0,
                        AntChunkStrides cs = AntChunkStrides{0, 0, 0, 0});
// per-chunk strides (doubles) of the round's row outputs / the injected action tape, for launches whose rows sit at different chunks
struct ChunkStrides { int64_t states, actions_out, actions_in; };
void launch_compact_ready(const int32_t* status, const int32_t* chunks_run, const int32_t* budget, int n_chunks, int B,
                          int32_t* idx_out, int32_t* nrow_out, int32_t* count, hipStream_t s);
int denoise_wave_quantum(ditree_ctx* ctx);
void launch_compact_alive(const int32_t* status, int B, int32_t* idx_out, int32_t* count, hipStream_t s,
                          const int32_t* budget = nullptr, int next_chunk = 0);
void launch_chunk_budget(const int32_t* parent, int B, const int32_t* num_visit, const int32_t* chunks, int n,
                         int32_t* budget, hipStream_t s);
void launch_gather_rows_f32(const float* src, int64_t src_stride, const int32_t* idx, float* dst, int row_floats, int n,
                            hipStream_t s);
void launch_car_rollout(const unsigned char* maze, int rows, int cols, double* state_io,
                        const double* actions, int64_t act_stride, int32_t* status_io, int B, int A,
                        double gx, double gy, double* states_out, int64_t states_stride,
                        double* actions_out, int64_t actout_stride, int32_t* steps_out,
                        double* prev_action_io, uint8_t* has_prev_io, hipStream_t s);
void launch_lidar_scan(const double* poses, int B, const float* maze, int rows, int cols, double* dist,
                       double* endpoints, uint8_t* hit, uint8_t* visited, hipStream_t s);
void launch_round_begin(int32_t* status, int32_t* chunks_run, int32_t* chunk_steps, int B, int n_chunks,
                        hipStream_t s);
void launch_round_chunk_end(const int32_t* chunk_status_in, int32_t* status, int32_t* chunks_run,
                            const double* cur_state, double* end_state, int B, hipStream_t s);
void launch_round_pack(const ditree_tree& t, const ditree_round& r, double* rec, hipStream_t s);
void launch_round_unpack(const ditree_tree& t, const ditree_round& r, const double* rec, hipStream_t s);
int record_doubles(const ditree_tree& t);
struct AheadArg { double t[30]; };
void launch_accept(const ditree_tree& t, const ditree_round& r, int emulate_sticky, const unsigned char* maze, int rows,
                   int cols, const AheadArg& ts, hipStream_t s);
void launch_obstacle_ahead(const unsigned char* maze, int rows, int cols, const double* state, int stride, int B,
                           const AheadArg& ts, uint8_t* out, hipStream_t s);
void launch_fallback_select(const ditree_tree& t, int n_nodes, double gx, double gy, const double* path_dev, int P,
                            int32_t* out_node, hipStream_t s);
void launch_follow_plan(double* state_io, const float* actions, int n_actions, int action_idx, const float* path, int P,
                        float* known, const float* truth, float* scanned, unsigned char* known_codes, int rows, int cols,
                        double gx, double gy, double dt, double scan_time, double* executed, int32_t* result,
                        hipStream_t s);
