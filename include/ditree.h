/*
 * ditree.h -- C-ABI of the MI355X-native DiTree expansion engine (libditree_hip.so).
 *
 * The reference (JJKK1313/DiTreeOnlinePlanner) is pure Python and has no FFI; its
 * "plugin boundary" for the hot path is the Python object surface that
 * run_scenarios*.py uses.  This header is the boundary the
 * build defines underneath that surface: every entry point below replaces the
 * numpy / scipy / PyTorch call site cited next to it (paths relative to the
 * reference root).  The Python facades in ditreeonlineplanner_amd/ bind these
 * symbols with ctypes (INTEGRATION.md shows the stub) and keep the reference's class
 * and argument names.
 *
 * Conventions
 *   - One ditree_ctx per (process, device).  A ctx is not re-entrant.
 *   - Every pointer argument marked [dev] is a caller-owned DEVICE pointer
 *     (e.g. torch.Tensor.data_ptr()); [host] is host memory read before return.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work
 *     is enqueued asynchronously on it; nothing synchronises the device.
 *   - Return value: 0 = ok, negative = DITREE_E_*; ditree_last_error(ctx) gives text
 *     valid until the next call on that ctx.  No exceptions, no exit().
 *   - Layouts are the reference's: car states (.., 6) f64 = x, y, psi, v, D, delta; car actions (.., 2) f64 = dD, ddelta;
 *     ant states (.., 29) f64 = achieved_goal (x, y) | observation (z, qw, qx, qy, qz, 8 joints, 14 velocities)
 *     (planners/base_planner.py:298), ant actions (.., 8); mazes row-major (rows, cols) f32 in {0, 1}.
 */
#ifndef DITREE_H
#define DITREE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DITREE_VERSION 400          /* 0.4.0: runtime state / action width of the tree (car 6 / 2, ant 29 / 8), the ant round as a
                                       real expansion round (collision, goal, accept), ant_rollout + stand-in model, row strides */

#define DITREE_OK 0
#define DITREE_E_ARG (-1)           /* bad argument (null pointer, size out of range) */
#define DITREE_E_HIP (-2)           /* a HIP runtime call failed */
#define DITREE_E_STATE (-3)         /* call order (no maze / no weights uploaded) */
#define DITREE_E_NOMEM (-4)

/* per-chunk / per-candidate status codes written by the rollout kernels */
#define DITREE_ST_NOT_RUN (-1)
#define DITREE_ST_OK 0              /* chunk finished: neither goal nor collision */
#define DITREE_ST_GOAL 1            /* planners/base_planner.py:300,314-317  (done is True) */
#define DITREE_ST_COLLIDED 2        /* planners/base_planner.py:306-312      (done is None) */
#define DITREE_ST_FLAG_GOAL_AT_COLLISION 0x100  /* the colliding step was inside the goal radius */

typedef struct ditree_ctx ditree_ctx;

int32_t ditree_version(void);
/* 16 hex digits: sha256 over the library's own sources (the csrc directory, this header) and compile flags, embedded at build time by
 * ditreeonlineplanner_amd/build.py; the Python binding refuses a library whose id differs from the sources next to it. */
const char* ditree_build_id(void);
int32_t ditree_ctx_create(int32_t device, ditree_ctx** out);
void ditree_ctx_destroy(ditree_ctx* ctx);
const char* ditree_last_error(ditree_ctx* ctx);

/* The planner's *known* occupancy grid (planners/base_planner.py:118 `self.maze`,
 * planners/RRT.py:57-59 update_maze).  Copies rows*cols floats host->device on
 * `stream`; used by ditree_local_map / ditree_car_rollout / ditree_expand_round. */
int32_t ditree_upload_maze(ditree_ctx* ctx, const float* maze /*[host] rows*cols*/,
                           int32_t rows, int32_t cols, void* stream);

/* planners/RRT.py:49-51 nearest_node (scipy KDTree.query, k = 1, first two dims) plus
 * the gather of the chosen node's fields (RRT.py:139-146).
 *   queries   [dev] (B, q_stride) f64, xy in columns 0..1
 *   node_xy   [dev] (N, 2) f64
 * Outputs (any may be NULL except out_idx):
 *   out_idx   [dev] (B,) i32  argmin_n ||q - node_n||^2, ties -> lowest n
 *   when node_state != NULL: out_state (B,6) = node_state[idx],
 *   out_prev_action (B,2) = node_last_action[idx], out_has_prev (B,) u8. */
int32_t ditree_nn_argmin(ditree_ctx* ctx, const double* queries, int32_t q_stride, int32_t B,
                         const double* node_xy, int32_t N, int32_t* out_idx,
                         const double* node_state, const double* node_last_action,
                         const uint8_t* node_has_prev, double* out_state,
                         double* out_prev_action, uint8_t* out_has_prev, void* stream);

/* common/map_utils.py:391-459 create_local_map on the uploaded maze.
 *   state [dev] (B, state_stride) f64, elements 0, 1, 2 used as x, y, psi (car: stride 6; the reference passes
 *   curr_state[2] for every env, planners/RRT.py:158-166, so the ant -- stride 29 -- rotates by its torso height);
 *   active [dev] (B,) i32 or NULL: rows with
 *   active[b] != DITREE_ST_OK are skipped;  axis [host] n doubles = the reference's
 *   np.linspace(-L/2 + s/2, L/2 - s/2, n) (map_utils.py:422-423), uploaded into ctx.
 *   out [dev] (B, n, n) f32; values m, or 2m-1 when scaled != 0 (policies/fm_policy.py:152). */
int32_t ditree_local_map(ditree_ctx* ctx, const double* state, int32_t state_stride, const int32_t* active,
                         int32_t B, int32_t n, const double* axis, double s_global, int32_t scaled,
                         float* out, void* stream);

/* policies/fm_policy.py:60-143, antmaze branch (obs_history 3, action_history 1, run_scenarios.py:123-132):
 * conditioning vector (B, 97) f32 = 3 x [z, rot6d(normalised quaternion) (6), joints + velocities (22)] | previous action
 * (8, normalised; raw zeros when has_prev == 0) | tanh((goal - position) / local_map_size) (2, yaw = 0).
 *   obs [dev] (B, n_hist, 29) f64, 1 <= n_hist <= 3 (missing steps are zero rows in front, fm_policy.py:96-102);
 *   prev_action [dev] (B, 8) f64; has_prev [dev] (B,) u8; cond_goal [dev] (B, 2) f64;
 *   norm [host] 70 doubles: obs_mean[27], obs_std[27], act_mean[8], act_std[8] (metadata/antmaze.pt). */
int32_t ditree_cond_vector_ant(ditree_ctx* ctx, const double* obs, int32_t n_hist, const double* prev_action,
                               const uint8_t* has_prev, const double* cond_goal, int32_t B, const double* norm,
                               double local_map_size, float* out, void* stream);

/* policies/fm_policy.py:60-143 (car): conditioning vector (B, 7) f32 =
 * [(v-5)/5, (D-.5)/.5, delta/.4 | (a_prev-mu)/sigma or 0,0 | tanh(R(-psi)(g-p)/lm_size)].
 *   norm [host] 16 doubles: obs_mean[6], obs_std[6], act_mean[2], act_std[2]. */
int32_t ditree_cond_vector(ditree_ctx* ctx, const double* state, const double* prev_action,
                           const uint8_t* has_prev, const double* cond_goal, int32_t B,
                           const double* norm, double local_map_size, float* out, void* stream);

/* planners/base_planner.py:257-320 propagate_action_sequence_env for B candidates:
 * A Euler steps of car_env.py:356-396, each followed by the goal test
 * (car_env.py:341-354) and the two-ball collision test (common/map_utils.py:103-115,
 * :221-329) against the uploaded maze.
 *   state_io   [dev] (B,6) f64  in: start state, out: end state (`obs`)
 *   actions    [dev] (B, act_stride) f64, rows of 2; the first A rows are used
 *   status_io  [dev] (B,) i32   in: rows != DITREE_ST_OK are skipped; out: chunk result
 *   states_out [dev] (B, states_stride) f64: (A+1, 6) rows; rows after the last
 *              executed step stay zero (base_planner.py:282)
 *   actions_out[dev] (B, actout_stride) f64: (A, 2) copy with rows after the goal step
 *              zeroed (base_planner.py:314-317)
 *   steps_out  [dev] (B,) i32 env steps executed (the colliding step counts)
 *   prev_action_io / has_prev_io: on DITREE_ST_OK updated to the chunk's last action
 *              (planners/RRT.py:188), may be NULL. */
int32_t ditree_car_rollout(ditree_ctx* ctx, double* state_io, const double* actions,
                           int64_t act_stride, int32_t* status_io, int32_t B, int32_t A,
                           const double* goal_xy /*[host] 2*/, double* states_out,
                           int64_t states_stride, double* actions_out, int64_t actout_stride,
                           int32_t* steps_out, double* prev_action_io, uint8_t* has_prev_io,
                           void* stream);

/* Where element (candidate b, row i, component k) of a per-candidate row block lives: base + b * cand + i * row + k * comp
 * doubles.  Rows packed per candidate (the reference's arrays): {rows * width, width, 1}.  Step-major, component-major,
 * candidate-minor -- {1, width * B, B} -- makes every store of a wavefront (64 consecutive candidates) one contiguous
 * 512-byte run instead of 64 fragments at a stride of rows * width * 8 bytes; what the large standalone rollouts use. */
typedef struct { int64_t cand, row, comp; } ditree_strides;

/* ditree_car_rollout with explicit layouts for the two row outputs (NULL = packed per candidate, i.e. ditree_car_rollout). */
int32_t ditree_car_rollout_ld(ditree_ctx* ctx, double* state_io, const double* actions, int64_t act_stride,
                              int32_t* status_io, int32_t B, int32_t A, const double* goal_xy /*[host] 2*/,
                              double* states_out, const ditree_strides* states_ld, double* actions_out,
                              const ditree_strides* actions_ld, int32_t* steps_out, double* prev_action_io,
                              uint8_t* has_prev_io, void* stream);

/* ------------------------------------------------------------------ ant: collision glue, stand-in dynamics, rollout */

/* common/map_utils.py:126-136 is_colliding_ant(state, maze, ant_radius, map_scale) as planners/base_planner.py:154-155 calls
 * it (1.2, s_global): upside down when R[2][2] = 1 - 2 (qx^2 + qy^2) < 0 with (qw, qx, qy, qz) = state[3:7]
 * (common/se3_utils.py:155-164), else the single-ball is_colliding_maze (common/map_utils.py:139-219: out of the map ->
 * collision; side walls when the ball crosses the cell's edge, beyond the map counting as a wall; corner cells when closer
 * than the radius, cells outside the map skipped) on the uploaded maze.
 *   state [dev] (B, stride >= 7) f64; out [dev] (B,) u8. */
int32_t ditree_ant_collision(ditree_ctx* ctx, const double* state, int32_t stride, int32_t B, double ball_radius,
                             double s_global, uint8_t* out, void* stream);

/* The ant's env step is MuJoCo 3.1.6 behind gymnasium-robotics 1.3.1 AntMaze_Large-v4 (requirements.txt:59,117; call sites
 * planners/base_planner.py:278-279,290): third party, not in the reference repository, no oracle -- NOT implemented.
 * What stands in for it in the higher-DoF rollout slot, PARITY UNPINNED BY CONSTRUCTION and labelled so wherever it runs:
 * this build's own surrogate of a four-legged crawler with the ant's interface (29-d observation, 8-d action clipped to
 * [-1, 1], frame_skip sub-steps of h seconds): per leg a hip and an ankle as damped second-order actuators with soft stops;
 * a loaded foot (load = (1 + tanh(contact_gain (ankle - ank_rest))) / 2) swept by its hip pushes the torso the other way
 * (planar force + yaw torque); uneven foot lift tilts the torso against an uprighting torque; the height follows the mean
 * lift; the quaternion integrates the body angular velocity and is re-normalised.  Defined by DESIGN.md and restated in
 * numpy as oracle/ant.py ant_model_step (tests hold the kernel to it at 1e-9). */
typedef struct {
  double h;                          /* integration step [s] (0.01) */
  double frame_skip;                 /* sub-steps per env step (5) */
  double k_act, k_spr, k_dmp, k_lim; /* actuator gain, joint spring, joint damping, soft-stop stiffness */
  double hip_lim, ank_lo, ank_hi, ank_rest;
  double contact_gain, leg_r, k_push, c_lin;
  double z0, z_gain, k_z, c_z;
  double k_lift, c_ang, k_up, k_yaw;
  double cphi, sphi;                 /* cos / sin of the first leg's mount angle (the others by symmetry) */
} ditree_ant_model;

/* planners/base_planner.py:257-320 for B ant candidates: A env steps, after each the goal test
 * ||obs[:2] - desired_goal|| < goal_radius (:296-297, goal_radius = 0.45 * s_global) and check_collision (:154-155,306;
 * ditree_ant_collision); collision wins, on goal the remaining action rows are zeroed (:315), unexecuted state rows stay 0.
 * The env step itself is `model` (the stand-in above) when model != NULL, else row i of next_obs_tape (test / measurement
 * infrastructure: the observation after every env step given from outside; with the real simulator the caller steps it on
 * the host between ditree_ant_chunk_sample and ditree_ant_chunk_step).
 *   state_io [dev] (B, 29) f64 in: chunk start, out: `obs`; actions [dev] (B, act_stride) f64 rows of 8;
 *   next_obs_tape [dev] (B, tape_stride) f64 rows of 29 or NULL; status_io as ditree_car_rollout;
 *   states_out [dev] A + 1 rows of 29 per candidate, actions_out [dev] A rows of 8, laid out by states_ld / actions_ld
 *   (NULL = packed per candidate); steps_out [dev] (B,) i32. */
int32_t ditree_ant_rollout(ditree_ctx* ctx, const ditree_ant_model* model, double* state_io, const double* actions,
                           int64_t act_stride, const double* next_obs_tape, int64_t tape_stride, int32_t* status_io,
                           int32_t B, int32_t A, const double* desired_goal_xy /*[host] 2*/, double goal_radius,
                           double ball_radius, double s_global, double* states_out, const ditree_strides* states_ld,
                           double* actions_out, const ditree_strides* actions_ld, int32_t* steps_out, void* stream);

/* lidar_sim/lidar_2d_sim.py:18-98 Lidar2DSim.scan for B poses (pose = x_col, y_row, yaw
 * in cell units) against `maze` [dev] (rows, cols) f32 (the *true* world, which may
 * differ from the uploaded known maze).  181 rays, arange(-180, 182, 2) degrees.
 *   dist [dev] (B,181) f64; endpoints [dev] (B,181,2) f64; hit [dev] (B,181) u8;
 *   visited [dev] (B, rows*cols) u8 or NULL: 1 where a ray sample fell before its hit. */
int32_t ditree_lidar_scan(ditree_ctx* ctx, const double* poses, int32_t B, const float* maze,
                          int32_t rows, int32_t cols, double* dist, double* endpoints,
                          uint8_t* hit, uint8_t* visited, void* stream);

/* ------------------------------------------------------------------ tree + rounds */

/* Device-resident tree (planners/base_planner.py:24-34 Node as SoA; all [dev]). */
typedef struct {
  int32_t capacity;          /* node slots */
  int32_t n_chunks, A;       /* edge_length / action_horizon, action_horizon */
  int32_t state_dim, action_dim;   /* S, D: 6 / 2 (car), 29 / 8 (ant); every "(.., 6)" / "(.., 2)" below reads (.., S) / (.., D) */
  double* state;             /* (cap, 6) */
  double* xy;                /* (cap, 2)  copy of state[:, :2] for the NN scan */
  int32_t* parent;           /* (cap,) */
  double* last_action;       /* (cap, 2)  last row of parent_action_seq */
  uint8_t* has_prev;         /* (cap,)    parent_action_seq is not None and not empty */
  int32_t* num_visit;        /* (cap,) */
  double* edge_states;       /* (cap, n_chunks*(A+1), 6)  zero-row-filtered, RRT.py:198-199 */
  double* edge_actions;      /* (cap, n_chunks*A, 2)      zero-row-filtered, RRT.py:196-197 */
  int32_t* edge_nstates;     /* (cap,) rows kept */
  int32_t* edge_nactions;    /* (cap,) */
  uint8_t* obstacle_ahead;   /* (cap,) or NULL: planners/RRT.py:61-81 flag of every appended node (run_type > 0) */
  int32_t* edge_owner;       /* (cap,) or NULL: rank that holds the node's edge rows (sharded rounds keep the trajectories
                                on the producing rank); -1 = every rank (root, frozen-env edges) */
  double* hist;              /* (cap, 3, S) or NULL: the last <= 3 rows of the node's filtered edge states, valid rows at the
                                END -- what the first sampler call of a child sees with obs_history 3 (planners/RRT.py:146-147,
                                policies/fm_policy.py:96-102; the ant).  Replicated on every rank (it travels in the record). */
  int32_t* hist_n;           /* (cap,) valid rows of hist (root: 1 = the start state) */
  int32_t* counters;         /* [0] n_nodes, [1] goal node (-1 none), [2] env.done latched
                                (car_env.py:254,266 "sticky done"), [3] chunk iterations,
                                [4] candidates processed, [5] sticky-done triggered */
} ditree_tree;

/* Per-round candidate records (all [dev]); B rows, written by ditree_expand_round or
 * by the individual ops, consumed by ditree_accept. */
typedef struct {
  int32_t B;
  int32_t* parent;           /* (B,) */
  int32_t* status;           /* (B,) final DITREE_ST_* (| FLAG_GOAL_AT_COLLISION) */
  int32_t* chunks_run;       /* (B,) */
  double* end_state;         /* (B, 6) */
  double* states;            /* (B, n_chunks, A+1, 6) */
  double* actions;           /* (B, n_chunks, A, 2) */
  int32_t* chunk_steps;      /* (B, n_chunks) */
  int32_t* node_id;          /* (B,) out: assigned node index or -1 */
  /* Sharded rounds (one process per GPU): only rows [own_lo, own_lo + own_n) of states / actions / chunk_steps were
   * produced here; the other rows arrive as 96-byte records (ditree_round_pack / _unpack), whose last / first action
   * land in the two arrays below.  Single rank: own_lo = 0, own_n = B, both arrays NULL. */
  double* last_action;       /* (B, 2) or NULL: last kept action row of the candidate's edge */
  double* first_action;      /* (B, 2) or NULL: actions[b, 0, 0, :] (the frozen-env edge of the sticky-done emulation) */
  int32_t own_lo, own_n;
  int32_t shard;             /* candidates per rank: owner of candidate b = b / shard (0 = single rank) */
  double* hist;              /* (B, 3, S) or NULL: trees with `hist`, sharded rounds: the exchanged end-of-edge history */
  int32_t* hist_n;           /* (B,) */
} ditree_round;

/* Candidate record exchanged between ranks once per round: S + 2 D + 2 doubles -- car: 12 = 96 bytes --
 * [end_state S | last_action D | first_action D | (parent, status) as two i32 | (chunks_run, hist_n) as two i32], followed by
 * [hist 3 S] for trees with `hist` (ant: 29 + 16 + 2 + 87 = 134 doubles).  ditree_record_doubles gives the count.
 * Edge trajectories stay on the producing rank.  pack: rows of `round` (this rank's slice) -> records_out (round->B, R);
 * unpack: records (B, R) of ALL candidates -> parent / status / chunks_run / end_state / last_action / first_action [/ hist]. */
#define DITREE_RECORD_DOUBLES 12    /* the car's record */
int32_t ditree_record_doubles(const ditree_tree* tree);
int32_t ditree_round_pack(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round, double* records_out,
                          void* stream);
int32_t ditree_round_unpack(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round, const double* records,
                            void* stream);

/* The per-round exchange for hosts without torch.distributed: an RCCL communicator inside the ctx (librccl is opened at
 * run time; one process per GPU, ranks of ONE node over xGMI).
 *   ditree_comm_unique_id: rank 0 fills a 128-byte id and hands it to the other ranks (file, pipe, MPI ...);
 *   ditree_comm_init:      every rank, collective;
 *   ditree_allgather_nodes: recv (world * count_doubles) <- every rank's send (count_doubles), in rank order, on `stream`
 *                           (in place when send == recv + rank * count_doubles). */
int32_t ditree_comm_unique_id(ditree_ctx* ctx, uint8_t* id128);
int32_t ditree_comm_init(ditree_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id128);
int32_t ditree_allgather_nodes(ditree_ctx* ctx, const double* send, double* recv, int64_t count_doubles, void* stream);
int32_t ditree_comm_destroy(ditree_ctx* ctx);

/* planners/RRT.py:179-217: accept candidates in index order (collided edges dropped,
 * lowest goal-reaching accepted index ends the plan, later candidates dropped),
 * filter all-zero rows, append nodes.  emulate_sticky != 0 reproduces the latched
 * env.done defect described in DESIGN.md. */
int32_t ditree_accept(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                      int32_t emulate_sticky, void* stream);

/* planners/RRT.py:61-81 check_obstacle_ahead on the uploaded maze: 30 samples over 1.5 cells ahead of the
 * heading in (col, row) space, int-truncated and clipped, any occupied.
 *   state [dev] (B, stride) f64 (x, y, psi used); out [dev] (B,) u8.  The sample offsets are
 *   np.linspace(0, 1.5, 30) = i * (1.5 / 29), last = 1.5, evaluated in f64 as numpy does. */
int32_t ditree_obstacle_ahead(ditree_ctx* ctx, const double* state, int32_t stride, int32_t B,
                              uint8_t* out, void* stream);

/* planners/RRT.py:83-111 extract_path_after_obstacle on the uploaded (known) maze: out2[0] = index c of the path point
 * nearest to cur_xy (first occurrence; np.linalg.norm(env.state[:2] - path[:, :2], axis=1) in the dtype numpy gives the
 * difference: f32_state != 0 when env.state is float32 -- right after env.reset, car_env.py:215 -- else float64 with the f32
 * path promoted), out2[1] = k such that the remaining reference path is path[c:][k:] -- the points behind the first blocked
 * stretch, cells looked up in the path's float32 (k = -1: the path crosses no obstacle, the reference then keeps its last
 * point only; k = P - c: nothing remains).  path [dev] (P, stride >= 2) f32 rows x, y, ...; cur_xy [host] 2 f64
 * (env.state[:2]); out2 [dev] 2 x i32. */
int32_t ditree_path_after_obstacle(ditree_ctx* ctx, const float* path, int32_t stride, int32_t P, const double* cur_xy /*[host] 2*/,
                                   int32_t f32_state, int32_t* out2, void* stream);

/* planners/RRT.py:233-254 fallback node when the budget ends without reaching the goal, over nodes
 * 1..n-1 of the tree: path == NULL: argmin ||xy - goal|| + 1e4 * obstacle_ahead; path != NULL
 * ([host] (P, 2) f64 = init_main_path xy): argmax of the nearest path index (-1 when an obstacle is
 * ahead).  First occurrence wins, as np.argmin / np.argmax.  out_node [dev] i32 (tree index), or -1 when
 * every node has an obstacle ahead or the tree holds only the start (RRT.py:227-232: no plan). */
int32_t ditree_fallback_select(ditree_ctx* ctx, const ditree_tree* tree, int32_t n_nodes,
                               const double* goal_xy /*[host] 2*/, const double* path /*[host] or NULL*/,
                               int32_t P, int32_t* out_node, void* stream);

/* Plan following of the online driver (run_scenarios_with_lidar_DiTree.py:470-506, run_type < 4) fused into
 * one launch: per action one env step with goal test and collision test against the KNOWN maze
 * (planners/base_planner.py:257-320 with a single action); whenever the accumulated step time exceeds
 * scan_time, scan_and_update_maze (:112-127: lidar scan of the TRUE maze from the pose, ray end cells written
 * as occupied into the known and the scanned maze, visited cells = 2) and check_no_obstacles_in_path
 * (:158-181, evaluated in the path's float32).  Stops at the first event.
 *   state_io [dev] (6) f64 in/out; actions [dev] (n_actions, 2) f32 (the plan's dtype); path_xy [dev] (P, 2) f32;
 *   known / scanned [dev] (rows, cols) f32 in/out, true_maze [dev] (rows, cols) f32 (rows, cols = the uploaded
 *   maze's; the ctx's own copy of the known maze is refreshed, so no re-upload is needed before the next
 *   expansion round); goal_xy [host] 2; executed [dev] (n_actions - action_idx, 6) f64: state after every
 *   executed action; result [dev] 4 x i32: event (DITREE_EV_*), next action index, first blocked path index
 *   or -1, number of scans.  Ray end cells outside the map (only possible without border walls, where the
 *   reference raises IndexError) are dropped.  rows * cols <= 13104 (five byte maps of the maze stay in LDS). */
#define DITREE_EV_ACTIONS_DONE 0
#define DITREE_EV_GOAL 1
#define DITREE_EV_COLLISION 2
#define DITREE_EV_OBSTACLE 3
int32_t ditree_follow_plan(ditree_ctx* ctx, double* state_io, const float* actions, int32_t n_actions,
                           int32_t action_idx, const float* path_xy, int32_t P, float* known_maze,
                           const float* true_maze, float* scanned_maze, const double* goal_xy /*[host] 2*/,
                           double dt, double scan_time, double* executed, int32_t* result, void* stream);

/* ------------------------------------------------------------------ MPPI controller */

/* One step of the MPPI controller `run_scenarios_with_lidar_MPPI.py:10,339-449` drives (`MPPI.mppi.MPPI(maze_data, T, K, nx,
 * nu)`: .reset / .step / .is_done / .set_ref_path / .reference_path / .update_maze / .env).  The reference repository does
 * not ship that module, so the algorithm is this build's own (DESIGN.md "MPPI"; PARITY UNPINNED): information-theoretic MPPI
 * on the reference's car dynamics (car_env.py:356-396), two-ball collision test (common/map_utils.py:103-115) and goal
 * radius (car_env.py:341-354), all against the maze uploaded with ditree_upload_maze.
 *   rollout k, step t:  u = U[t] + eps[k, t] (rollout 0: eps = 0), x <- car_step(x, u);
 *     cost += w_track * d2(x, path) + lambda * sum_d U[t, d] eps[k, t, d] / sigma_d^2;  d2 = squared distance to the nearest
 *     path point with index in [i - window_back, i + window_fwd] around the previous step's nearest index i (the first step
 *     starts from the nearest index of the current state over the whole path);
 *     collision: cost += w_collision, rollout ends;  goal radius: cost -= w_goal, rollout ends;
 *     terminal: cost += w_progress * (P - 1 - i_last).
 *   beta = min_k S_k;  w_k = exp(-(S_k - beta) / lambda);  U[t] += sum_k w_k eps[k, t] / sum_k w_k.
 *   execute: a = clip(U[0]) -> one env step; collision: state unchanged, U <- 0, status 2; else state <- x', U shifted by
 *   one step (last control held), status 1 inside the goal radius, else 0.
 * stages (bit mask, DITREE_MPPI_*): ROLLOUTS (fills costs [, flags]), MIN (beta = min_k S_k -> result[3]), SUMS (w_k from
 * result[3]; sums [dev] (3 + 2T) = {eta = sum w, sum w^2, collided rollouts, sum_k w_k eps[k, t, d]}), APPLY (U += sums[3..] /
 * sums[0], weights normalised, result[4], [6], [7]), EXECUTE (env step + shift).  DITREE_MPPI_ALL = one controller step.
 * SHARDED over ranks (BASELINE config 5: 65 536 rollouts over 8 GPUs; one process per GPU): every rank runs ROLLOUTS | MIN
 * on its K rollouts with k_offset = its first GLOBAL rollout index, the ranks all-reduce result[3] (MIN), run SUMS,
 * all-reduce `sums` (SUM), then APPLY | EXECUTE -- two collectives of 1 and 3 + 2T doubles per controller step; state and
 * controls stay replicated.  noise [dev] (K, T, 2) f64 or NULL = generated on the device, a pure function of (seed,
 * counter, GLOBAL k, t) (splitmix64 -> Box-Muller) that never touches HBM.
 *   state_io [dev] 6 f64; U_io [dev] (T, 2) f64 nominal controls; path_xy [dev] (P, 2) f64, P <= 4096; goal_xy [host] 2;
 *   costs [dev] (K) f64; weights [dev] (K) f64 or NULL (normalised w_k); flags [dev] (K) i32 or NULL (0, 1 goal, 2 collided);
 *   result [dev] 16 f64: executed action (2), status, beta, eta = sum_k w_k, nearest path index of the input state, number of
 *   collided rollouts (needs flags), effective sample size eta^2 / sum w^2, [8..13] the state after EXECUTE (one D2H per step). */
#define DITREE_MPPI_ROLLOUTS 1
#define DITREE_MPPI_MIN 2
#define DITREE_MPPI_SUMS 4
#define DITREE_MPPI_APPLY 8
#define DITREE_MPPI_EXECUTE 16
#define DITREE_MPPI_ALL 31
typedef struct {
  int32_t T, K;                      /* horizon (<= 64), rollouts */
  double lambda;                     /* temperature */
  double sigma[2];                   /* noise standard deviation of (dD, ddelta) */
  double w_track, w_progress, w_collision, w_goal;
  uint64_t seed;
  int32_t window_back, window_fwd;   /* nearest-path-index search window per step */
  int32_t lanes;                     /* lanes of a wavefront that share one rollout: 1, 2 or 4 (one ball and a share of the path
                                        window per lane: that many waves per SIMD at K = 65 536), 0 = the measured default.
                                        Same results whatever the value. */
  int64_t k_offset;                  /* global index of this rank's first rollout (0 for a single rank) */
} ditree_mppi_params;
int32_t ditree_mppi_step(ditree_ctx* ctx, const ditree_mppi_params* p, double* state_io, double* U_io, const double* path_xy,
                         int32_t P, const double* goal_xy, const double* noise, uint64_t counter, int32_t stages,
                         double* costs, double* weights, int32_t* flags, double* sums, double* result, void* stream);

/* The same controller on the higher-DoF rollout slot -- BASELINE config 5 as written ("MPPI antmaze, 65 536 rollouts"): 29-d
 * state, 8-d controls U (T, 8) with noise eps ~ N(0, diag(sigma^2)), every rollout step one env step of the build's STAND-IN
 * model (ditree_ant_model; NOT MuJoCo), the reference's ant collision test (ditree_ant_collision) and goal radius, the same
 * cost, soft-min update, executed step and stages as ditree_mppi_step.  The reference ships neither an MPPI module nor the
 * ant's physics: PARITY UNPINNED BY CONSTRUCTION (the build's numpy restatement: oracle/mppi.py).
 *   state_io [dev] 29 f64; U_io [dev] (T, 8); path_xy [dev] (P, 2); noise [dev] (K, T, 8) or NULL (counter hash, four
 *   Box-Muller pairs per (GLOBAL rollout, step)); costs / weights / flags as ditree_mppi_step; sums [dev] 3 + 8 T;
 *   result [dev] 64 f64: [2] status, [3] beta, [4] eta, [5] nearest path index of the input state, [6] collided rollouts,
 *   [7] effective sample size, [16..24) the executed action, [24..53) the state after EXECUTE. */
typedef struct {
  int32_t T, K;
  double lambda;
  double sigma[8];
  double w_track, w_progress, w_collision, w_goal;
  uint64_t seed;
  int32_t window_back, window_fwd;
  int64_t k_offset;
  double goal_radius, ball_radius, s_global;     /* 0.45 * s_global, 1.2, 4 (planners/base_planner.py:155,297) */
  ditree_ant_model model;
} ditree_mppi_ant_params;
int32_t ditree_mppi_step_ant(ditree_ctx* ctx, const ditree_mppi_ant_params* p, double* state_io, double* U_io,
                             const double* path_xy, int32_t P, const double* desired_goal_xy /*[host] 2*/, const double* noise,
                             uint64_t counter, int32_t stages, double* costs, double* weights, int32_t* flags, double* sums,
                             double* result, void* stream);

/* ------------------------------------------------------------------ denoiser */

/* Upload the denoiser weights (reference: run_scenarios.py:157-185, state-dict keys
 * `encoder.resnet18.*`, `unet.*`).  `blob` [host] is the flat fp32 parameter blob and
 * `manifest` [host] a NUL-terminated text table "name offset n_elems dims...\n" built by
 * ditreeonlineplanner_amd/weights.py; the library repacks into MFMA-friendly bf16
 * (and, for the f32 parity instantiation, fp32) tiles on the device. */
int32_t ditree_load_weights(ditree_ctx* ctx, const float* blob, int64_t n_floats,
                            const char* manifest, void* stream);

/* Allocate the activation workspace for up to max_batch candidates. */
int32_t ditree_denoise_reserve(ditree_ctx* ctx, int32_t max_batch, int32_t precision);
#define DITREE_PREC_BF16 0          /* bf16 MFMA inputs, fp32 accumulate (throughput path; 8 significand bits) */
#define DITREE_PREC_F32 1           /* fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's arithmetic, 1/16 of the bf16 rate */
#define DITREE_PREC_F16X3 2         /* f32-class: every operand as f16 hi + lo planes, 3 f16 MFMAs per product (22 bits),
                                       f32 accumulate, the encoder split the same way; 1/3 of the 16-bit rate.  f16 range:
                                       activations beyond +-65504 are clamped AND reported (ditree_denoise_status) */
#define DITREE_PREC_BF16X3 3        /* the same split on bf16 (16 significand bits, f32 range) */
#define DITREE_PREC_F16 4           /* plain f16 MFMA inputs (11 significand bits) at the bf16 rate */

/* Range guard of the f16 instantiations (DITREE_PREC_F16X3 / _F16).  The reference computes in fp32
 * (model/diffusion/conditional_unet1d.py:110-117: `scale * out + bias` of a FiLM layer has no bound); f16 activations
 * saturate at +-65504.  Every layer that stores f16 activations raises a sticky device flag when it is handed a value
 * beyond that range.  This call WAITS for `stream` (the one exception to "nothing synchronises"), reads the flags and
 * returns the number of layers that saturated since the last clear; their state-dict names (e.g.
 * "unet.mid_modules.0.blocks.0.block.0"), newline-separated and NUL-terminated, go to names [host] (names_cap bytes, may
 * be 0).  clear != 0 resets the flags.  A result > 0 means the actions of those calls are NOT the network's: re-reserve
 * with DITREE_PREC_BF16X3 (f32 exponent range) or DITREE_PREC_F32.  Always 0 for the bf16 / f32 instantiations. */
int32_t ditree_denoise_status(ditree_ctx* ctx, int32_t* n_saturated /*[host]*/, char* names /*[host]*/, int64_t names_cap,
                              int32_t clear, void* stream);

/* Dimensions of the loaded denoiser: dims5 = {pred_horizon, action_dim, local_map_size, obs-cond width, map embedding}. */
int32_t ditree_denoise_dims(ditree_ctx* ctx, int32_t* dims5);

/* policies/fm_policy.py:155-203 + local_map_encoder.py:101-109 +
 * model/diffusion/conditional_unet1d.py:268-347: K flow steps of
 * x <- x + net(x, map, 20*t0[k], cond) * dt[k], then a = x*sigma + mu.
 *   noise     [dev] (B, P, 2) f32   x0 ~ N(0, I)
 *   local_map [dev] (B, 20, 20) f32 already scaled to {-1, +1}
 *   cond      [dev] (B, 7) f32
 *   t0, dt    [host] K floats (common/fm_utils.py:4-17)
 *   act_norm  [host] 2*D doubles mu[D], sigma[D] (D = action_dim of the loaded net)
 *   actions   [dev] (B, P, 2) f64   un-normalised actions
 *   x_out     [dev] (B, P, 2) f32 or NULL: the normalised sample after the last step */
int32_t ditree_denoise(ditree_ctx* ctx, const float* noise, const float* local_map,
                       const float* cond, int32_t B, int32_t K, const float* t0, const float* dt,
                       const double* act_norm, double* actions, float* x_out, void* stream);

/* The sampler's policy = 'diffusion' branch (policies/fm_policy.py:164-182) with a DDPM scheduler as the reference configures it
 * (run_scenarios.py:157-158: beta_schedule 'squaredcos_cap_v2', clip_sample True, prediction_type 'epsilon'; variance
 * 'fixed_small'), K reverse steps inside the library:
 *   eps = net(x, map, timesteps[k], cond)            (the embedding sees the scheduler's integer timestep, unscaled)
 *   x0  = clip((x - sb_k eps) / sa_k, -1, 1);   x <- c0_k x0 + c1_k x + sigma_k z_k
 * with, for timestep t and its predecessor (abar = cumulative product of 1 - beta, abar_prev = 1 behind the last step):
 *   sb = sqrt(1 - abar_t), sa = sqrt(abar_t), c0 = sqrt(abar_prev) (1 - abar_t / abar_prev) / (1 - abar_t),
 *   c1 = sqrt(abar_t / abar_prev) (1 - abar_prev) / (1 - abar_t), sigma = sqrt(max((1 - abar_prev) / (1 - abar_t) (1 - abar_t /
 *   abar_prev), 1e-20)) for t > 0 and 0 at t = 0 -- all formed by the caller in float32 as the scheduler's tensors are.
 * `diffusers` is not part of the reference repository: the arithmetic follows the published algorithm (Ho et al. 2020 as
 * diffusers 0.x implements it) and is held to the build's numpy restatement -- PARITY UNPINNED.
 *   noise [dev] (B, P, D) f32 x_K ~ N(0, I); step_noise [dev] (B, K, P, D) f32 z_k ~ N(0, I) (row k unused where sigma_k = 0);
 *   timesteps [host] K floats; coef [host] (K, 5) floats sb, sa, c0, c1, sigma; other arguments as ditree_denoise. */
int32_t ditree_denoise_ddpm(ditree_ctx* ctx, const float* noise, const float* step_noise, const float* local_map,
                            const float* cond, int32_t B, int32_t K, const float* timesteps, const float* coef,
                            const double* act_norm, double* actions, float* x_out, void* stream);

/* One raw evaluation of the noise-prediction network, out = net(sample, map, timestep, cond)
 * (model/diffusion/conditional_unet1d.py:268-347 behind local_map_encoder.py:101-109): the building block of the
 * sampler's policy = 'diffusion' branch (policies/fm_policy.py:164-182), whose scheduler step stays with the caller.
 *   sample [dev] (B, P, 2) f32; timestep: the value the sinusoidal embedding sees (the scheduler's k);
 *   reuse_encoder != 0 keeps the map embedding of the previous call (same local maps, later steps of one loop);
 *   out [dev] (B, P, 2) f32. */
int32_t ditree_denoise_eval(ditree_ctx* ctx, const float* sample, const float* local_map, const float* cond,
                            int32_t B, float timestep, int32_t reuse_encoder, float* out, void* stream);

/* Measurement support.  enable = 1: every launch of the MFMA kernels is bracketed by hipEvents on its launch
 * stream (back-to-back launches share an event; ~120 marker packets per denoiser call, which costs a timed
 * region ~6 %); enable = 2: only the dominant kernel, one event pair around each run of back-to-back halo
 * launches (~20 packets per call); 0: off.  ditree_profile_read waits for the events and returns, per kernel
 * kind k = 0 conv3_halo16_kernel, 1 gemm16_kernel / conv_gemm_kernel, 2 conv2d_small_kernel: summed kernel time ms3[k],
 * launches3[k] and executed FLOPs flops3[k] (2*M*N*K of every launch) since the last enable. */
int32_t ditree_profile(ditree_ctx* ctx, int32_t enable);
int32_t ditree_profile_read(ditree_ctx* ctx, double* ms3, int64_t* launches3, double* flops3);

/* Test / debug support: copy a named internal activation of the last ditree_denoise call
 * (e.g. "d0b1.out", "skip2", "mid2.out", "final.h", "film", "map_emb") as f32
 * (B, L, C) into out [dev]; dims3 [host] receives (B, L, C). */
int32_t ditree_denoise_debug_read(ditree_ctx* ctx, const char* name, int32_t B, float* out,
                                  int64_t capacity, int32_t* dims3, void* stream);

/* One expansion round for B candidates (planners/RRT.py:131-194 batched):
 * nearest node -> n_chunks x [local map -> cond -> denoise (or injected actions)
 * -> rollout].  Fills `round`; does not modify the tree (call ditree_accept, after
 * the cross-rank all-gather when sharded). */
typedef struct {
  int32_t n_nodes;               /* tree size to search */
  const double* samples;         /* [dev] (B, 6) */
  const double* cond_goal;       /* [dev] (B, 2) */
  const float* noise;            /* [dev] (B, n_chunks, P, 2) f32, or NULL with inject_actions */
  const double* inject_actions;  /* [dev] (B, n_chunks, P, 2) f64 or NULL: bypass the denoiser */
  int32_t P;                     /* pred_horizon */
  int32_t K;                     /* flow steps */
  const float* t0;               /* [host] K */
  const float* dt;               /* [host] K */
  const double* norm;            /* [host] 16 doubles, see ditree_cond_vector */
  const double* goal_xy;         /* [host] 2: env.goal = centre of the goal cell */
  const double* axis;            /* [host] local-map axis, n doubles */
  int32_t lm_n;                  /* local_map_size */
  double lm_size;                /* the divisor of the goal conditioning (local_map_size) */
  double s_global;
  int32_t early_exit;            /* != 0: chunks run for the candidates that are still alive only (RRT.py:179-184), the denoiser calls
                                    packed to whole tile-waves from the pool of ready (candidate, chunk) items; same results */
  const float* ddpm_coef;        /* [host] (K, 5) or NULL.  != NULL: the sampler's DDPM branch (ditree_denoise_ddpm) instead of the
                                    flow steps; t0 then holds the K scheduler timesteps and dt is unused */
  const float* step_noise;       /* [dev] (B, n_chunks, K, P, 2) f32 with ddpm_coef */
  const int32_t* chunk_budget;   /* [dev] (B,) chunks each candidate may run (ditree_chunk_budget), or NULL = all n_chunks */
} ditree_round_params;

/* planners/RRT.py:26,149-152: `edge_length = prop_duration[clip(curr_node.num_visit)]; curr_node.num_visit += 1`.
 * For a round expanded against one tree snapshot the candidates of a parent are its visits in candidate order:
 * budget[b] = schedule[clip(num_visit[parent(b)] + #{b' < b : parent(b') == parent(b)})], in chunks of action_horizon steps.
 *   samples [dev] (B, 6): ALL candidates of the round (every rank computes the whole round's budgets: the rank inside a
 *   parent's visit order is a global property); schedule_chunks [host] n_schedule ints, each <= tree->n_chunks;
 *   parent_scratch [dev] (B,) i32 workspace (the nearest nodes); budget_out [dev] (B,) i32. */
int32_t ditree_chunk_budget(ditree_ctx* ctx, const ditree_tree* tree, const double* samples, int32_t B, int32_t n_nodes,
                            const int32_t* schedule_chunks, int32_t n_schedule, int32_t* parent_scratch,
                            int32_t* budget_out, void* stream);

int32_t ditree_expand_round(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                            const ditree_round_params* p, void* stream);

/* Scheduling figures of the last ditree_expand_round with early_exit on this ctx: stats4 [host] = {denoiser calls, tile-waves
 * they cost (sum over calls of ceil(rows / quantum)), the quantum = candidates per full wave of tiles on the 256 CUs for the
 * loaded denoiser (512 for the car network), 0}.  An early-exit round packs its calls to whole waves from a pool of ready
 * (candidate, next chunk) items (DESIGN.md "Early exit"); without early exit a round of B candidates and n chunks costs
 * n * ceil(B / quantum) waves. */
int32_t ditree_round_stats(ditree_ctx* ctx, int32_t* stats4);

/* One expansion round of the ANT (BASELINE config 3; cfgs/antmaze.yaml + run_scenarios.py:123-132: action_horizon 2, edge
 * length 48 = 24 chunks, pred_horizon 16, obs_history 3, local map 16 x 16 @ 0.8, s_global 4) against a tree with state_dim
 * 29, action_dim 8 and `hist`: planners/RRT.py:131-194 batched --
 *   nearest node (RRT.py:49-51) -> its state, last action and the last <= 3 rows of its edge (RRT.py:144-147) ->
 *   n_chunks x [create_local_map at (state[0], state[1], state[2]) (RRT.py:158-166: element 2 -- the torso height -- is what
 *   the reference passes as the rotation) -> ant conditioning vector (policies/fm_policy.py:77-143: normalise, quaternion ->
 *   rot6d of the normalised values, 3-step history, previous action, tanh of the goal offset with yaw 0) -> denoiser (input_dim 8)
 *   -> the first A un-normalised actions -> A env steps with goal + collision test (ditree_ant_rollout) ->
 *   prev_states = curr_states_seq, prev_actions = curr_action_seq (RRT.py:186-190)].
 * Fills `round` (status / chunks_run / chunk_steps / states (B, n_chunks, A + 1, 29) / actions (B, n_chunks, A, 8) /
 * end_state); ditree_accept then appends the nodes (emulate_sticky = 0: the ant env has no latched `done`).
 * THE ENV STEP (MuJoCo, not in the reference repository, no oracle) is one of:
 *   DITREE_ANT_DYN_TAPE   next_obs_tape (B, n_chunks, A, 29): observations given from outside (tests, measurement);
 *   DITREE_ANT_DYN_MODEL  the build's stand-in model (ditree_ant_model; parity unpinned, not MuJoCo);
 *   the caller's own simulator, one chunk at a time: ditree_ant_round_begin, then per chunk ditree_ant_chunk_sample (actions of
 *   the chunk -> round->actions[:, j], start states in round->end_state), the host steps its env, ditree_ant_chunk_step with
 *   the observations.  ditree_expand_round_ant is exactly that sequence with the tape / model in the middle. */
#define DITREE_ANT_DYN_TAPE 0
#define DITREE_ANT_DYN_MODEL 1
typedef struct {
  int32_t n_nodes;               /* tree size to search */
  const double* samples;         /* [dev] (B, 29): columns 0, 1 used (the KD-tree is over x, y, RRT.py:23,50) */
  const double* cond_goal;       /* [dev] (B, 2) */
  const float* noise;            /* [dev] (B, n_chunks, P, 8) f32, or NULL with inject_actions */
  const double* inject_actions;  /* [dev] (B, n_chunks, P, 8) f64 or NULL: bypass the denoiser (tests: action tapes) */
  int32_t P;                     /* pred_horizon (16) */
  int32_t K;                     /* flow steps */
  const float* t0;               /* [host] K */
  const float* dt;               /* [host] K */
  const double* norm;            /* [host] 70: obs_mean[27], obs_std[27], act_mean[8], act_std[8] (metadata/antmaze.pt) */
  const double* desired_goal;    /* [host] 2: obs['desired_goal'] of the env (base_planner.py:296) */
  double goal_radius;            /* 0.45 * s_global */
  double ball_radius;            /* 1.2 (base_planner.py:155) */
  const double* axis;            /* [host] local-map axis, lm_n doubles */
  int32_t lm_n;                  /* local_map_size (16) */
  double lm_size;                /* divisor of the goal conditioning */
  double s_global;               /* 4 */
  int32_t dynamics;              /* DITREE_ANT_DYN_* (ditree_expand_round_ant only) */
  const double* next_obs_tape;   /* [dev] (B, n_chunks, A, 29) f64 for DITREE_ANT_DYN_TAPE */
  const ditree_ant_model* model; /* [host] for DITREE_ANT_DYN_MODEL */
  int32_t early_exit;            /* != 0: later chunks run on the still-alive candidates only (RRT.py:179-184) */
  const float* ddpm_coef;        /* [host] (K, 5) or NULL: the DDPM branch, as ditree_round_params */
  const float* step_noise;       /* [dev] (B, n_chunks, K, P, 8) f32 with ddpm_coef */
  float* cond_out;               /* [dev] (B, n_chunks, 97) f32 or NULL: the conditioning vector of every sampler call (tests) */
} ditree_ant_round_params;
int32_t ditree_expand_round_ant(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                                const ditree_ant_round_params* p, void* stream);
/* The same round with the caller's simulator between the two halves of a chunk (j = 0 .. n_chunks - 1, in order):
 *   begin:  status / counters reset, nearest node, parent state -> round->end_state, history + previous action -> ctx;
 *   sample: local map, conditioning, denoiser for the candidates that are still alive; round->actions[:, j] <- the A actions;
 *   step:   next_obs [dev] (B, A, 29) f64 = the observation after each of the A env steps from round->end_state[b] with
 *           round->actions[b, j] (rows of candidates that are not alive are ignored; a candidate that collides or reaches the
 *           goal at step i ignores the rows after i) -> tests, round->states[:, j], status, history for the next chunk. */
int32_t ditree_ant_round_begin(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                               const ditree_ant_round_params* p, void* stream);
int32_t ditree_ant_chunk_sample(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                                const ditree_ant_round_params* p, int32_t j, void* stream);
int32_t ditree_ant_chunk_step(ditree_ctx* ctx, const ditree_tree* tree, const ditree_round* round,
                              const ditree_ant_round_params* p, int32_t j, const double* next_obs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DITREE_H */
