// Internal interface between the denoiser's host orchestration and its kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { MODE_BIAS = 0, MODE_GN_MISH = 1, MODE_GN_MISH_FILM = 2, MODE_GN_MISH_RES = 3 };

// Numeric format of an activation buffer / a GEMM: storage type | split << 2.
//   storage: bf16, f32, f16.  split (16-bit storage only): the buffer holds two planes, hi = rnd16(x) and
//   lo = rnd16(x - hi), `plane` bytes apart; a split GEMM forms hi*hi + hi*lo + lo*hi on the 16-bit MFMA pipeline.
enum { ST_BF16 = 0, ST_F32 = 1, ST_F16 = 2 };
static inline int fmt_make(int st, bool split) { return st | (split ? 4 : 0); }
static inline __host__ __device__ int fmt_st(int f) { return f & 3; }
static inline __host__ __device__ bool fmt_split(int f) { return (f & 4) != 0; }
static inline int fmt_es(int f) { return fmt_st(f) == ST_F32 ? 4 : 2; }

// One implicit-GEMM launch (see denoise_kernels.hip).  All counts are in elements of the
// activation type (bf16 or f32); pointers are device pointers.
struct ConvGemmParams {
  const void* A;      // activations, channels contiguous
  int lda;            // elements per A row
  int in_Lp;          // A rows per sample (L_in + 2 for padded conv inputs)
  int in_stride;      // row(b, l, tap) = b*in_Lp + l*in_stride + tap + in_off
  int in_off;
  int taps, Cin;      // K = taps * Cin, Cin a multiple of 64
  const void* W;      // [ceil(N/256)*256][taps*Cin], row co = [tap][ci]
  void* Out;
  int ldc;            // elements per output row
  int out_Lp, out_stride, out_off;   // out row(b, l) = b*out_Lp + l*out_stride + out_off
  int out_coff;       // channel offset inside the output row (concat halves)
  int L;              // positions per sample in this GEMM; M = samples * L
  int M, N;
  const float* bias;  // [N] or null
  int mode;
  const float* gamma; // [N] GroupNorm affine
  const float* beta;
  int group_ch;       // channels per group (N / 8)
  float eps;
  const float* film;  // [samples][film_ld] f32: scale at film_off + c, bias at film_off + N + c
  int film_ld, film_off;
  const void* Res;    // residual, activation type
  int ldres, res_Lp, res_off;
  int out_f32;        // store f32 regardless of the activation type
  int dbg;            // halo / gemm16 kernels: 1 = walk an XCD's tile range in strips of four tile columns (xcd_remap_strips)
  // implicit Conv2d on NHWC activations (c2d != 0): GEMM row m = (b, oh, ow); K = live taps x Cin
  // (Cin a multiple of 64: one tap spans Cin/64 K-steps); A is the input activation (B, H, W, Cin);
  // taps that fall outside the map read `zero` (>= 128 zero bytes).
  int c2d, c2_H, c2_W, c2_OW, c2_OHW, c2_stride, c2_pad;
  signed char c2_kh[12], c2_kw[12];
  const void* zero;
  // split formats: bytes from the hi plane to the lo plane of each operand; f16 weights are stored multiplied by a power
  // of two (the range of f16 is narrow) and `w_scale` = its reciprocal is applied to the accumulator
  long long a_plane, w_plane, out_plane, res_plane;
  float w_scale;
  // split-K (implicit Conv2d mode only): gridDim.y = splitk blocks share a tile, block y handles a
  // contiguous range of K-steps and writes its partial f32 tile to Out + y * slab_stride elements
  int splitk;
  long long slab_stride;
  // split-K of the halo kernel (latency mode, small batches): gridDim.y = splitk work-groups share a tile, each walks nv / splitk
  // channel chunks and dumps its f32 accumulators to sk_ws[(y * tiles + tile) * 65536]; the last one to arrive (sk_cnt[tile])
  // adds the slabs in split order -- one fixed order whoever arrives last -- and runs the epilogue
  float* sk_ws;
  int* sk_cnt;
  // f16 range guard: device flag word of this layer (set to 1 by any thread that stores a value beyond +-65504; null = off).
  // Only the f16 instantiations look at it.
  int* sat;
};

void denoise_set_dry_run(bool on);   // launchers check their contracts but enqueue nothing (plan validation at reserve time)
void launch_conv_gemm(const ConvGemmParams& p, int fmt, hipStream_t s);
bool conv2d_small_eligible(int fmt);                    // implicit Conv2d layers run on the 64 x 64-tile kernel
int conv_gemm_kind(const ConvGemmParams& p, int fmt);   // 0 halo kernel, 1 gemm16 / generic, 2 implicit Conv2d
bool conv_gemm_supported(const ConvGemmParams& p, int fmt);   // split formats: only shapes of the halo / gemm16 tiles
void launch_prep_sample(const float* x, void* A0, int B, int P, int D, int fmt, long long plane, hipStream_t s, int* sat = nullptr);
void launch_time_embed(float t, const float* W1, const float* b1, const float* W2, const float* b2, float* out,
                       hipStream_t s);
void launch_prep_cond(const float* temb, const float* map_emb, int E, int E_ld, const float* cond, int G, void* out, int B,
                      int Kpad, int fmt, long long plane, hipStream_t s, int* sat = nullptr);
// What the last projection does with the network output v of a sampler step:
//   FLOW  x += v * dt                     (policies/fm_policy.py:183-196)
//   RAW   x  = v                          (one raw network evaluation)
//   DDPM  x0 = clip((x - sb v) / sa, -1, 1); x = c0 x0 + c1 x + sigma z   (policies/fm_policy.py:164-182 with a DDPM scheduler:
//         epsilon prediction, clip_sample, fixed_small variance; z = the step's standard-normal noise, row r of it for dense row r
//         = z[(z_idx ? z_idx[r] : z_row0 + r) * z_row + (l * D + d)]; sigma == 0 at the last step)
struct FlowStep {
  int mode;                 // 0 FLOW, 1 RAW, 2 DDPM
  float dt;
  float sb, sa, c0, c1, sigma;
  const float* z;
  long long z_row;
  const int* z_idx;
  int z_row0;
};
void launch_final_proj_flow(const void* Y, int C, int Lp, long long plane, const float* W, const float* bias, int D, float* x,
                            FlowStep fs, const double* act_norm /*[mu[D], sigma[D]]*/, double* actions, int B, int P, int fmt,
                            hipStream_t s);
struct TapList {          // live (kh, kw) taps of a Conv2d on a small map
  int n;
  signed char kh[49], kw[49];
};
void launch_gn1d(void* x, int ld, int Lp, int row_off, int coff, int L, int C, const float* gamma, const float* beta, float eps,
                 int mode, const float* film, int film_ld, int film_off, const void* res, int ldres, int res_Lp, int res_off,
                 int B, int fmt, long long x_plane, long long res_plane, hipStream_t s, int* sat = nullptr);
void launch_encoder_stem(const float* lm, int n, const float* W, const float* gamma, const float* beta, void* out, int B, float eps,
                         int fmt, long long plane, hipStream_t s, int* sat = nullptr);
void launch_im2col2d(const void* in, bool src_f32, void* out, int B, int H, int W, int C, const TapList& taps, int stride,
                     int pad, int OH, int OW, int Kpad, int fmt, hipStream_t s);
void launch_gn2d(const float* in, int nslab, long long slab_stride, const float* gamma, const float* beta, const void* res,
                 int relu, void* out, int B, int HW, int C, float eps, int fmt, long long res_plane, long long out_plane,
                 hipStream_t s, int* sat = nullptr);
void launch_maxpool2d(const void* in, void* out, int B, int H, int W, int C, int OH, int OW, int fmt, hipStream_t s);
void launch_avgpool2d(const void* in, void* out, int B, int HW, int C, int fmt, long long in_plane, long long out_plane,
                      hipStream_t s);
void launch_unpack_act(const void* in, int ld, int coff, int Lp, int roff, float* out, int B, int L, int C, int fmt,
                       long long plane, hipStream_t s);
