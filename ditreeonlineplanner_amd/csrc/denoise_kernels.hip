// MFMA kernels of the DiTree denoiser (gfx950 / CDNA4).
//
// One implicit-GEMM kernel serves every dense layer of the reference network
// (model/diffusion/conditional_unet1d.py:41-142,268-347, conv1d_components.py:7-40,
//  local_map_encoder.py:112-122):
//      Out[(b,l), co] = sum_{tap, ci} X[b, l*stride + tap + off, ci] * W[co, tap, ci]
//  * activations are channels-last with one zero row on either side of every sample
//    ([B][L+2][C]), so the k = 3 taps of a Conv1d are three *overlapping row windows* of
//    the same buffer -- no im2col copy, the A operand is streamed straight from HBM/L2
//    into LDS with `global_load_lds` (16 B per lane) and per-lane source addresses;
//  * GEMM orientation: M = positions (B*L), N = output channels, K = taps*C_in;
//    256 x 256 x (128 bytes of K) tiles, 8 waves as 4(M) x 2(N), each wave 64 x 128 with
//    v_mfma_f32_32x32x16_bf16 (PREC 0) or v_mfma_f32_32x32x2_f32 (PREC 1, parity path);
//  * LDS rows are 128 B; the 16-B slot index is XOR-swizzled with (row>>1)&7 on the
//    *source* address and on the ds_read_b128 address (conflict-free fragment reads);
//  * the weight rows of a tile are permuted in LDS so that lane r of a wave owns output
//    channels 4r..4r+3: the epilogue then stores 8 B (bf16) / 16 B (f32) per lane,
//    256 / 512 contiguous bytes per output row;
//  * epilogue fused in registers: bias, GroupNorm (two-pass statistics through LDS
//    atomics; a tile always holds whole (sample, group) sets), Mish, FiLM scale/bias or
//    residual add.
#include "denoise.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN preserved
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
  unsigned int x = ((unsigned int)u) << 16;
  return __builtin_bit_cast(float, x);
}

template <int PREC>
__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)) = x * w / (w + 2), w = e^x (e^x + 2); softplus threshold 20 as torch
  if (x > 20.0f) return x;
  float n = (PREC == 0) ? __expf(x) : expf(x);
  float w = n * (n + 2.0f);
  return x * (w / (w + 2.0f));
}

// bijective XCD-aware tile remap (blocks b and b+8 share an XCD): each XCD gets a
// contiguous range of tiles, so concurrently resident tiles share A / W panels in L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
  int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + k;
}

template <int PREC>
__global__ void __launch_bounds__(512, 2) conv_gemm_kernel(ConvGemmParams p) {
  constexpr int ES = (PREC == 0) ? 2 : 4;          // element bytes
  constexpr int EK = 128 / ES;                     // elements of K per step
  constexpr int EPS = 16 / ES;                     // elements per 16-B slot
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r5 = lane & 31, h = lane >> 5;
  const int wm = w >> 1, wn = w & 1;
  const int ntn = (p.N + 255) >> 8;
  const int ntm = (p.M + 255) >> 8;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int K = p.taps * p.Cin;
  const int nk = K / EK;
  const int kpt = p.Cin / EK;                      // K-steps per tap

  // ---- per-lane staging sources (byte offsets) ------------------------------------------
  const char* Abase = (const char*)p.A;
  const char* Wbase = (const char*)p.W;
  long long aoff[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (w * 4 + q) * 8 + (lane >> 3);          // LDS row 0..255
    const int sw = (r >> 1) & 7;
    const int slot = (lane & 7) ^ sw;
    int m = tm * 256 + r;
    m = m < p.M ? m : p.M - 1;
    const int b = m / p.L, l = m - b * p.L;
    aoff[q] = (((long long)b * p.in_Lp + (long long)l * p.in_stride + p.in_off) * p.lda + slot * EPS) * ES;
    // weight row permutation: LDS row rho = wn*128 + j*32 + rr  <->  channel wn*128 + 4*rr + j
    const int c = (r & 128) + 4 * (r & 31) + ((r >> 5) & 3);
    boff[q] = (((long long)(tn * 256 + c)) * K + slot * EPS) * ES;
  }
  auto stage = [&](int kt, int buf) {
    const int tap = kt / kpt;
    const long long akoff = ((long long)tap * p.lda + (long long)(kt - tap * kpt) * EK) * ES;
    const long long bkoff = (long long)kt * EK * ES;
    char* sbase = smem + buf * 65536;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(Abase + aoff[q] + akoff),
                                       (LDS_AS void*)(sbase + (w * 4 + q) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(Wbase + boff[q] + bkoff),
                                       (LDS_AS void*)(sbase + 32768 + (w * 4 + q) * 1024), 16, 0, 0);
    }
  };

  f32x16_t acc[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][j][i] = 0.0f;

  const int swl = (r5 >> 1) & 7;
  const int a_row_off = (wm * 64 + r5) * 128;               // + mb*32*128
  const int b_row_off = 32768 + (wn * 128 + r5) * 128;      // + j*32*128

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* sb = smem + (kt & 1) * 65536;
    if constexpr (PREC == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ps = (((ks << 1) | h) ^ swl) << 4;
        bf16x8_t af[2], bfr[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          af[mb] = __builtin_bit_cast(bf16x8_t, *(const short8_t*)(sb + a_row_off + mb * 4096 + ps));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bfr[j] = __builtin_bit_cast(bf16x8_t, *(const short8_t*)(sb + b_row_off + j * 4096 + ps));
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[j], acc[mb][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {                       // 16-B slot = 4 floats = k 4*sl .. 4*sl+3
        const int ps = (sl ^ swl) << 4;
        f32x4_t af[2], bfr[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) af[mb] = *(const f32x4_t*)(sb + a_row_off + mb * 4096 + ps);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const f32x4_t*)(sb + b_row_off + j * 4096 + ps);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {                     // k pair (2*s2, 2*s2+1): lane half h takes k = 2*s2 + h
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float a = h ? af[mb][2 * s2 + 1] : af[mb][2 * s2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float bv = h ? bfr[j][2 * s2 + 1] : bfr[j][2 * s2];
              acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[mb][j], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  __syncthreads();                                            // all fragment reads done: LDS reusable

  // ---- epilogue ---------------------------------------------------------------------------
  const int c_l = wn * 128 + 4 * r5;                          // lane's 4 consecutive channels in the tile
  const int n0 = tn * 256 + c_l;
  const bool n_ok = n0 < p.N;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && n_ok) {
    f32x4_t t = *(const f32x4_t*)(p.bias + n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = t[j];
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][j][i] += bias4[j];

  // 16-row blocks of this lane: blk = mb*2 + hf covers tile rows wm*64 + mb*32 + hf*16 .. +15
  int blk_b[4], blk_l[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const int m0 = tm * 256 + wm * 64 + (blk >> 1) * 32 + (blk & 1) * 16;
    const int b = m0 / p.L;
    blk_b[blk] = b;
    blk_l[blk] = m0 - b * p.L;
  }

  if (p.mode >= MODE_GN_MISH) {
    float* s_sum = (float*)smem;                              // [16 slots][4 groups]
    float* s_sq = s_sum + 64;
    if (tid < 128) s_sum[tid] = 0.0f;
    __syncthreads();
    const int spt = 256 / p.L;                                // sample slots per tile
    const int gi = c_l / p.group_ch;                          // lane's group within the tile
    const bool wide = p.group_ch >= 128;
    int slot[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) slot[blk] = blk_b[blk] - tm * spt;
    const float inv_cnt = 1.0f / (float)(p.group_ch * p.L);
    float mean[4], rstd[4];
    // pass 1: sums
    {
      float ps[4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) s += acc[blk >> 1][j][(blk & 1) * 8 + i];
        ps[blk] = s;
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        float s = ps[blk];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 32);
        if (wide) s += __shfl_xor(s, 16);
        if (lane == 0 || (!wide && lane == 16)) atomicAdd(&s_sum[slot[blk] * 4 + gi], s);
      }
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) mean[blk] = s_sum[slot[blk] * 4 + gi] * inv_cnt;
    // pass 2: centred squares
    {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float d = acc[blk >> 1][j][(blk & 1) * 8 + i] - mean[blk];
            s += d * d;
          }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 32);
        if (wide) s += __shfl_xor(s, 16);
        if (lane == 0 || (!wide && lane == 16)) atomicAdd(&s_sq[slot[blk] * 4 + gi], s);
      }
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) rstd[blk] = rsqrtf(s_sq[slot[blk] * 4 + gi] * inv_cnt + p.eps);

    f32x4_t gam = *(const f32x4_t*)(p.gamma + n0);
    f32x4_t bet = *(const f32x4_t*)(p.beta + n0);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4_t fs = {1.f, 1.f, 1.f, 1.f}, fb = {0.f, 0.f, 0.f, 0.f};
      if (p.mode == MODE_GN_MISH_FILM) {
        const float* fr = p.film + (long long)blk_b[blk] * p.film_ld + p.film_off + n0;
        fs = *(const f32x4_t*)fr;
        fb = *(const f32x4_t*)(fr + p.N);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ga = gam[j] * rstd[blk], be = bet[j] - mean[blk] * ga;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v = acc[blk >> 1][j][(blk & 1) * 8 + i];
          v = mish_f<PREC>(v * ga + be);
          acc[blk >> 1][j][(blk & 1) * 8 + i] = v * fs[j] + fb[j];
        }
      }
    }
  }

  // ---- store (+ residual) --------------------------------------------------------------------
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ii = (blk & 1) * 8 + i;
      const int rofs = (i & 3) + 8 * (i >> 2) + 4 * h;         // row within the 16-row block
      const int m = tm * 256 + wm * 64 + (blk >> 1) * 32 + (blk & 1) * 16 + rofs;
      if (m >= p.M || !n_ok) continue;
      const int b = blk_b[blk], l = blk_l[blk] + rofs;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[blk >> 1][j][ii];
      if (p.mode == MODE_GN_MISH_RES) {
        const long long rrow = (long long)b * p.res_Lp + l + p.res_off;
        if constexpr (PREC == 0) {
          short4_t rv = *(const short4_t*)((const char*)p.Res + (rrow * p.ldres + n0) * 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += bf2f((unsigned short)rv[j]);
        } else {
          f32x4_t rv = *(const f32x4_t*)((const char*)p.Res + (rrow * p.ldres + n0) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += rv[j];
        }
      }
      const long long orow = (long long)b * p.out_Lp + (long long)l * p.out_stride + p.out_off;
      const long long oidx = orow * p.ldc + p.out_coff + n0;
      if (p.out_f32 || PREC == 1) {
        f32x4_t o = {v[0], v[1], v[2], v[3]};
        *(f32x4_t*)((char*)p.Out + oidx * 4) = o;
      } else {
        short4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (short)f2bf(v[j]);
        *(short4_t*)((char*)p.Out + oidx * 2) = o;
      }
    }
  }
}

void launch_conv_gemm(const ConvGemmParams& p, int prec, hipStream_t s) {
  const int ntn = (p.N + 255) >> 8, ntm = (p.M + 255) >> 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv_gemm_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)conv_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  if (prec == 0)
    hipLaunchKernelGGL(conv_gemm_kernel<0>, dim3(ntm * ntn), dim3(512), 131072, s, p);
  else
    hipLaunchKernelGGL(conv_gemm_kernel<1>, dim3(ntm * ntn), dim3(512), 131072, s, p);
}

// ============================================================================= small kernels
template <int PREC>
__device__ __forceinline__ void store_elem(void* base, long long idx, float v) {
  if constexpr (PREC == 0) ((unsigned short*)base)[idx] = f2bf(v);
  else ((float*)base)[idx] = v;
}
template <int PREC>
__device__ __forceinline__ float load_elem(const void* base, long long idx) {
  if constexpr (PREC == 0) return bf2f(((const unsigned short*)base)[idx]);
  else return ((const float*)base)[idx];
}

// x (B, P, D) f32 -> A0 rows (b, l): [x[l-1,:], x[l,:], x[l+1,:], 0 ...] (K padded to 64): the
// im2col of the first Conv1d(D -> C, 3) (conditional_unet1d.py:214-218 with dim_in = input_dim).
template <int PREC>
__global__ void prep_sample_kernel(const float* __restrict__ x, void* __restrict__ A0, int B, int P, int D) {
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * P) return;
  const int b = (int)(row / P), l = (int)(row - (long long)b * P);
  float v = 0.f;
  if (lane < 3 * D) {
    const int t = lane / D, d = lane - t * D;
    const int ls = l + t - 1;
    if (ls >= 0 && ls < P) v = x[((long long)b * P + ls) * D + d];
  }
  store_elem<PREC>(A0, row * 64 + lane, v);
}
void launch_prep_sample(const float* x, void* A0, int B, int P, int D, int prec, hipStream_t s) {
  long long rows = (long long)B * P;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (prec == 0) hipLaunchKernelGGL(prep_sample_kernel<0>, grid, block, 0, s, x, A0, B, P, D);
  else hipLaunchKernelGGL(prep_sample_kernel<1>, grid, block, 0, s, x, A0, B, P, D);
}

// Time embedding of one flow step: sinusoidal(256) -> Linear(256,1024) -> Mish -> Linear(1024,256)
// (positional_embedding.py:10-17, conditional_unet1d.py:180-185).  Batch-invariant, f32.
__global__ void __launch_bounds__(1024) time_embed_kernel(float t, const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          float* __restrict__ out /*[256]*/) {
  __shared__ float emb[256];
  __shared__ float hid[1024];
  const int tid = threadIdx.x;
  if (tid < 256) {
    const int half = 128;
    const float wlog = logf(10000.0f) / (float)(half - 1);
    const int k = tid & 127;
    const float f = expf((float)k * -wlog);
    const float a = t * f;
    emb[tid] = (tid < 128) ? sinf(a) : cosf(a);
  }
  __syncthreads();
  {
    float s = b1[tid];
    const float* wr = W1 + (long long)tid * 256;
    for (int k = 0; k < 256; ++k) s += wr[k] * emb[k];
    hid[tid] = mish_f<1>(s);
  }
  __syncthreads();
  if (tid < 256) {
    float s = b2[tid];
    const float* wr = W2 + (long long)tid * 1024;
    for (int k = 0; k < 1024; ++k) s += wr[k] * hid[k];
    out[tid] = s;
  }
}
void launch_time_embed(float t, const float* W1, const float* b1, const float* W2, const float* b2, float* out,
                       hipStream_t s) {
  hipLaunchKernelGGL(time_embed_kernel, dim3(1), dim3(1024), 0, s, t, W1, b1, W2, b2, out);
}

// FiLM input: Mish(cat(time_emb 256, map_emb E, obs_cond G)) zero-padded to Kpad columns
// (conditional_unet1d.py:59-64 cond_encoder = Mish -> Linear, :293 global_feature).
template <int PREC>
__global__ void prep_cond_kernel(const float* __restrict__ temb, const float* __restrict__ map_emb, int E,
                                 const float* __restrict__ cond, int G, void* __restrict__ out, int B, int Kpad) {
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    bool live = true;
    if (k < 256) v = temb[k];
    else if (k < 256 + E) v = map_emb[(long long)b * E + (k - 256)];
    else if (k < 256 + E + G) v = cond[(long long)b * G + (k - 256 - E)];
    else live = false;
    store_elem<PREC>(out, (long long)b * Kpad + k, live ? mish_f<PREC>(v) : 0.f);
  }
}
void launch_prep_cond(const float* temb, const float* map_emb, int E, const float* cond, int G, void* out, int B,
                      int Kpad, int prec, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(prep_cond_kernel<0>, dim3(B), dim3(256), 0, s, temb, map_emb, E, cond, G, out, B, Kpad);
  else hipLaunchKernelGGL(prep_cond_kernel<1>, dim3(B), dim3(256), 0, s, temb, map_emb, E, cond, G, out, B, Kpad);
}

// Final Conv1d(C -> D, 1) + flow Euler step + un-normalise (conditional_unet1d.py:253-256,
// policies/fm_policy.py:193,201-203).  One wave per position; Y is the padded channels-last
// output of the last Conv1dBlock.
template <int PREC>
__global__ void __launch_bounds__(256) final_proj_flow_kernel(const void* __restrict__ Y, int C, int Lp,
                                                              const float* __restrict__ W /*[D][C]*/,
                                                              const float* __restrict__ bias, int D,
                                                              float* __restrict__ x /*[B][P][D] in/out*/, float dt,
                                                              double mu0, double mu1, double sg0, double sg1,
                                                              double* __restrict__ actions, int B, int P) {
  const long long pos = blockIdx.x * 4LL + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pos >= (long long)B * P) return;
  const int b = (int)(pos / P), l = (int)(pos - (long long)b * P);
  const long long row = (long long)b * Lp + l + 1;
  float s[2] = {0.f, 0.f};
  for (int c = lane; c < C; c += 64) {
    const float y = load_elem<PREC>(Y, row * C + c);
    for (int d = 0; d < 2; ++d) s[d] += (d < D) ? y * W[(long long)d * C + c] : 0.f;
  }
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s[d] += __shfl_xor(s[d], m);
  }
  if (lane < D && lane < 2) {
    const float v = s[lane] + bias[lane];
    const long long xi = pos * D + lane;
    const float xn = x[xi] + v * dt;                        // naction + vel_pred * dt[k]
    x[xi] = xn;
    if (actions != nullptr) {
      const double sg = lane == 0 ? sg0 : sg1, mu = lane == 0 ? mu0 : mu1;
      actions[xi] = (double)xn * sg + mu;                   // float32 * float64 -> float64 (:203)
    }
  }
}
void launch_final_proj_flow(const void* Y, int C, int Lp, const float* W, const float* bias, int D, float* x, float dt,
                            const double* act_norm, double* actions, int B, int P, int prec, hipStream_t s) {
  long long pos = (long long)B * P;
  dim3 grid((unsigned)((pos + 3) / 4)), block(256);
  if (prec == 0)
    hipLaunchKernelGGL(final_proj_flow_kernel<0>, grid, block, 0, s, Y, C, Lp, W, bias, D, x, dt, act_norm[0],
                       act_norm[1], act_norm[2], act_norm[3], actions, B, P);
  else
    hipLaunchKernelGGL(final_proj_flow_kernel<1>, grid, block, 0, s, Y, C, Lp, W, bias, D, x, dt, act_norm[0],
                       act_norm[1], act_norm[2], act_norm[3], actions, B, P);
}

// ------------------------------------------------------------------------------- encoder helpers
// im2col for Conv2d on NHWC activations: out row (b, oh, ow), column (kh*KW + kw)*C + c, zero
// padded to Kpad.  SRC_F32: the source is the f32 local map (B, H, W) with C = 1.
template <int PREC, bool SRC_F32>
__global__ void im2col2d_kernel(const void* __restrict__ in, void* __restrict__ out, int B, int H, int W, int C,
                                int KH, int KW, int stride, int pad, int OH, int OW, int Kpad) {
  const long long row = blockIdx.x;
  const int ow = (int)(row % OW), oh = (int)((row / OW) % OH), b = (int)(row / ((long long)OW * OH));
  const int K = KH * KW * C;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < K) {
      const int c = k % C, kk = k / C, kw = kk % KW, kh = kk / KW;
      const int ih = oh * stride + kh - pad, iw = ow * stride + kw - pad;
      if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
        const long long idx = (((long long)b * H + ih) * W + iw) * C + c;
        v = SRC_F32 ? ((const float*)in)[idx] : load_elem<PREC>(in, idx);
      }
    }
    store_elem<PREC>(out, row * Kpad + k, v);
  }
}
void launch_im2col2d(const void* in, bool src_f32, void* out, int B, int H, int W, int C, int KH, int KW, int stride,
                     int pad, int OH, int OW, int Kpad, int prec, hipStream_t s) {
  dim3 grid((unsigned)((long long)B * OH * OW)), block(Kpad >= 256 ? 256 : 64);
  if (prec == 0) {
    if (src_f32) hipLaunchKernelGGL((im2col2d_kernel<0, true>), grid, block, 0, s, in, out, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad);
    else hipLaunchKernelGGL((im2col2d_kernel<0, false>), grid, block, 0, s, in, out, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad);
  } else {
    if (src_f32) hipLaunchKernelGGL((im2col2d_kernel<1, true>), grid, block, 0, s, in, out, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad);
    else hipLaunchKernelGGL((im2col2d_kernel<1, false>), grid, block, 0, s, in, out, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad);
  }
}

// GroupNorm (C/16 groups, local_map_encoder.py:63-76) on the f32 GEMM output [B][HW][C],
// optional residual add and ReLU (torchvision BasicBlock), writes the activation type.
template <int PREC>
__global__ void __launch_bounds__(64) gn2d_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const void* __restrict__ res,
                                                  int relu, void* __restrict__ out, int HW, int C, float eps) {
  const int groups = C >> 4;
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int lane = threadIdx.x;
  const int n = HW * 16;
  const long long base = (long long)b * HW * C + g * 16;
  float s = 0.f;
  for (int e = lane; e < n; e += 64) s += in[base + (long long)(e >> 4) * C + (e & 15)];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)n;
  float q = 0.f;
  for (int e = lane; e < n; e += 64) {
    float d = in[base + (long long)(e >> 4) * C + (e & 15)] - mean;
    q += d * d;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m);
  const float rstd = rsqrtf(q / (float)n + eps);
  for (int e = lane; e < n; e += 64) {
    const long long idx = base + (long long)(e >> 4) * C + (e & 15);
    const int c = g * 16 + (e & 15);
    float v = (in[idx] - mean) * rstd * gamma[c] + beta[c];
    if (res != nullptr) v += load_elem<PREC>(res, idx);
    if (relu) v = v > 0.f ? v : 0.f;
    store_elem<PREC>(out, idx, v);
  }
}
void launch_gn2d(const float* in, const float* gamma, const float* beta, const void* res, int relu, void* out, int B,
                 int HW, int C, float eps, int prec, hipStream_t s) {
  dim3 grid((unsigned)(B * (C >> 4))), block(64);
  if (prec == 0) hipLaunchKernelGGL(gn2d_kernel<0>, grid, block, 0, s, in, gamma, beta, res, relu, out, HW, C, eps);
  else hipLaunchKernelGGL(gn2d_kernel<1>, grid, block, 0, s, in, gamma, beta, res, relu, out, HW, C, eps);
}

// MaxPool2d(3, 2, 1) on NHWC.
template <int PREC>
__global__ void maxpool2d_kernel(const void* __restrict__ in, void* __restrict__ out, int B, int H, int W, int C, int OH,
                                 int OW) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long total = (long long)B * OH * OW * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int ow = (int)((idx / C) % OW), oh = (int)((idx / ((long long)C * OW)) % OH), b = (int)(idx / ((long long)C * OW * OH));
  float best = -__builtin_huge_valf();
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
      if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
        float v = load_elem<PREC>(in, (((long long)b * H + ih) * W + iw) * C + c);
        best = v > best ? v : best;
      }
    }
  store_elem<PREC>(out, idx, best);
}
void launch_maxpool2d(const void* in, void* out, int B, int H, int W, int C, int OH, int OW, int prec, hipStream_t s) {
  long long total = (long long)B * OH * OW * C;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == 0) hipLaunchKernelGGL(maxpool2d_kernel<0>, grid, block, 0, s, in, out, B, H, W, C, OH, OW);
  else hipLaunchKernelGGL(maxpool2d_kernel<1>, grid, block, 0, s, in, out, B, H, W, C, OH, OW);
}

// AdaptiveAvgPool2d(1) on NHWC -> [B][C].
template <int PREC>
__global__ void avgpool2d_kernel(const void* __restrict__ in, void* __restrict__ out, int B, int HW, int C) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)B * C) return;
  const int c = (int)(idx % C), b = (int)(idx / C);
  float s = 0.f;
  for (int q = 0; q < HW; ++q) s += load_elem<PREC>(in, ((long long)b * HW + q) * C + c);
  store_elem<PREC>(out, idx, s / (float)HW);
}
void launch_avgpool2d(const void* in, void* out, int B, int HW, int C, int prec, hipStream_t s) {
  long long total = (long long)B * C;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == 0) hipLaunchKernelGGL(avgpool2d_kernel<0>, grid, block, 0, s, in, out, B, HW, C);
  else hipLaunchKernelGGL(avgpool2d_kernel<1>, grid, block, 0, s, in, out, B, HW, C);
}

// debug / test support: padded channels-last activation -> f32 [B][L][C]
template <int PREC>
__global__ void unpack_act_kernel(const void* __restrict__ in, int ld, int coff, int Lp, int roff, float* __restrict__ out,
                                  int B, int L, int C) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)B * L * C) return;
  const int c = (int)(idx % C);
  const int l = (int)((idx / C) % L), b = (int)(idx / ((long long)C * L));
  out[idx] = load_elem<PREC>(in, ((long long)b * Lp + l + roff) * ld + coff + c);
}
void launch_unpack_act(const void* in, int ld, int coff, int Lp, int roff, float* out, int B, int L, int C, int prec,
                       hipStream_t s) {
  long long total = (long long)B * L * C;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == 0) hipLaunchKernelGGL(unpack_act_kernel<0>, grid, block, 0, s, in, ld, coff, Lp, roff, out, B, L, C);
  else hipLaunchKernelGGL(unpack_act_kernel<1>, grid, block, 0, s, in, ld, coff, Lp, roff, out, B, L, C);
}
