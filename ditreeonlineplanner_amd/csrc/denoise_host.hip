// Host side of the denoiser: weight repacking, activation workspace, launch sequence.
//
// Reference structure reproduced layer for layer:
//   local_map_encoder.py:101-122   encoder(local_map) -> 400-d, cat with obs_cond
//   conditional_unet1d.py:268-347  time MLP, down [CRB,CRB,Down] x3 (last Identity),
//                                  mid 2 x CRB, up [cat skip, CRB, CRB, Up] x2, final conv
//   policies/fm_policy.py:183-203  flow steps x <- x + v*dt[k]; a = x*sigma + mu
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <thread>

#include "denoise.h"
#include "ditree_internal.h"

namespace {

struct HostParam {
  std::vector<int64_t> dims;
  const float* data = nullptr;
  int64_t n = 0;
};

struct Act {             // channels-last activation view
  void* p = nullptr;
  int L = 0, C = 0;      // positions per sample, channels of this view
  int ld = 0, coff = 0;  // row stride (elements), channel offset of the view
  bool padded = true;    // rows per sample = L + 2 (one zero row each side)
  long long plane = 0;   // split formats: bytes from the hi plane to the lo plane (same layout)
  int fmt = 0;           // numeric format of the buffer (denoise.h)
  int Lp() const { return padded ? L + 2 : L; }
};

struct Packed {          // a GEMM weight in the MFMA layout
  void* p = nullptr;
  long long plane = 0;   // split formats: bytes to the lo plane
  float scale = 1.0f;    // f16: the stored values are w / scale (scale a power of two)
};

inline uint16_t f2bf_host(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf2f_host(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline uint16_t f2h_host(float f) { _Float16 h = (_Float16)f; uint16_t u; std::memcpy(&u, &h, 2); return u; }   // RNE
inline float h2f_host(uint16_t u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

}  // namespace

struct DenoiserState {
  ditree_ctx* ctx = nullptr;
  std::vector<float> blob;
  std::map<std::string, HostParam> params;
  // architecture (derived from the parameter shapes)
  int D = 2, P = 64, E = 400, G = 7, cond_dim = 663, lm = 20;
  int dims[3] = {512, 1024, 2048};
  bool loaded = false;
  // DDPM loop of the NEXT denoise call (set by denoise_run_ddpm around denoise_core): coef (K, 5) = sb, sa, c0, c1, sigma per step
  const float* ddpm_coef = nullptr;
  const float* ddpm_z = nullptr;           // [dev] step noise, row r / step k at r * z_row + k * z_step
  long long ddpm_z_row = 0, ddpm_z_step = 0;
  int ddpm_row0 = 0;                       // first row of the current sub-batch (dense calls)
  // workspace
  int prec = -1, Bmax = 0;     // prec: the DITREE_PREC_* the workspace was built for; Bmax: samples the WORKSPACE holds
  int Buser = 0;               // samples the caller reserved for: calls up to this size are served in sub-batches of <= Bmax
  int bgran = 16;              // batch rows are padded to a multiple of this
  int ufmt = 0, efmt = 0;      // formats (denoise.h) of the U-Net activations / GEMMs and of the encoder
  std::vector<void*> allocs;
  std::map<std::string, Act> named;
  std::map<std::string, Packed> dev_w;      // packed GEMM weights by name
  std::map<std::string, float*> dev_f;      // f32 vectors (bias, gamma, beta, small matrices)
  std::vector<std::function<void(int, int, hipStream_t)>> unet_ops;            // (B, Bp, stream)
  std::vector<std::function<void(int, int, int, hipStream_t)>> enc_ops;        // (b0, Bn, scratch region, stream)
  static constexpr int ENC_SUBS = 4;     // encoder sub-batches that may run concurrently
  int sub_cap = 0;
  size_t col_region = 0, gout_region = 0;
  hipStream_t aux[ENC_SUBS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[ENC_SUBS] = {nullptr, nullptr, nullptr, nullptr};
  std::function<void(int, int, hipStream_t)> film_op;
  float* x_cur = nullptr;        // (Bmax, P, D) f32
  float* temb = nullptr;         // (256,) f32
  float temb_t = -1.0f;          // timestep the cached embedding was computed for (< 0: none)
  float* map_emb = nullptr;      // (Bmax, E_ld) f32
  int E_ld = 400;                // row stride of map_emb (E, or E padded to 256 for the split encoder's fc tiles)
  float* film = nullptr;         // (Bp, film_cols) f32
  void* condA = nullptr;         // (Brows, condK) [x planes]
  long long cond_plane = 0;
  int condK = 0, film_cols = 0;
  Act final_h;                   // input of the final 1x1 projection
  const float* lm_ptr = nullptr; // caller's scaled local map of the current call
  void* zero_row = nullptr;      // 256 zero bytes: source of out-of-map taps in implicit Conv2d
  // f16 range guard (denoise_kernels.hip sat_*): one device flag word per layer that stores f16 activations; a kernel ORs 1
  // into its layer's word when it is handed a value beyond +-65504.  Sticky until ditree_denoise_status reads and clears.
  static constexpr int SAT_SLOTS = 256;
  int* sat_flags = nullptr;
  int *sat_cond = nullptr, *sat_sample = nullptr;      // slots of the two input-preparation kernels
  std::vector<std::string> sat_names;
  int* sat_slot(const std::string& layer, int fmt) {
    if (fmt_st(fmt) != ST_F16 || sat_flags == nullptr) return nullptr;
    for (size_t i = 0; i < sat_names.size(); ++i)
      if (sat_names[i] == layer) return sat_flags + i;
    if ((int)sat_names.size() >= SAT_SLOTS) return sat_flags + SAT_SLOTS - 1;
    sat_names.push_back(layer);
    return sat_flags + sat_names.size() - 1;
  }
  int last_splitk[4] = {1, 1, 1, 1};       // split-K layout of the conv output the next GroupNorm op reads, per region
  long long last_slab[4] = {0, 0, 0, 0};
  // optional per-launch timing of the dominant kernel (hipEvents on the launch stream)
  bool prof_on = false;
  int prof_mode = 0;             // 1: every MFMA launch, 2: only runs of the dominant (halo) kernel, one event pair per run
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  // per kernel kind: 0 = conv3_halo_kernel, 1 = conv_gemm_kernel, 2 = conv_gemm_kernel (implicit Conv2d)
  double prof_ms_done[3] = {0, 0, 0};
  int64_t prof_launches_done[3] = {0, 0, 0};
  double prof_flops_done[3] = {0, 0, 0};
  struct ProfRec { size_t a, b; int kind; double flops; int launches; };
  bool run_open = false;         // mode 2: a run of back-to-back halo launches whose end event is still to be recorded
  size_t run_end = 0;
  void close_run() {
    if (run_open) { hipEventRecord(prof_ev[run_end], prof_last_stream); run_open = false; }
  }
  // Profiling brackets every GEMM launch with events on its stream.  Back-to-back GEMM launches on
  // one stream share an event (the end of one is the start of the next), which halves the marker
  // packets; `prof_chain` is broken by note_other() whenever another kernel is enqueued in between.
  hipStream_t prof_last_stream = nullptr;
  bool prof_chain = false;
  std::vector<ProfRec> prof_pairs;                           // (start event, end event, kind, flops) per launch
  void note_other() { close_run(); prof_chain = false; }     // call BEFORE enqueueing the other kernel
  // Latency mode (DITREE_DENOISE_SPLITK=1, read per call; opt-in): below ~ 256 candidates a layer is a handful of 256 x 256 tiles
  // and the K loop of ONE tile (up to 384 K-steps) is the layer's latency while most CUs idle.  The 3-tap convs of the split
  // formats then run `splitk` work-groups per tile (each a contiguous share of the channel chunks), the last one adds the partial
  // accumulators in split order and runs the fused epilogue.  Deterministic, but NOT bit-identical to the one-pass sum: with
  // the mode on, a sample's result depends on whether its batch was small enough to split -- which is why it is opt-in and why
  // a sharded planner must switch it by its GLOBAL batch.
  float* sk_ws = nullptr;
  int* sk_cnt = nullptr;
  static constexpr int SK_MAX_SLABS = 512;                   // 512 x 256 KB = 128 MB of partial tiles
  int pick_splitk(const ConvGemmParams& p, int fmt) {
    const char* e = getenv("DITREE_DENOISE_SPLITK");
    if (!e || atoi(e) == 0 || !fmt_split(fmt) || p.L == 4 || conv_gemm_kind(p, fmt) != 0) return 1;   // (L = 4: no split-K form)
    const int tiles = (p.M >> 8) * (p.N >> 8), nvc = p.Cin >> 5;
    int sk = 1;
    // more work-groups only while the chip is mostly idle: every split adds a 256 KB partial tile to write and read back.
    // Measured (profiles/r04_splitk_latency_probe.json): a denoiser call of 1 / 16 / 64 / 128 samples 6.8 -> 4.9 / 4.9 / 5.0 / 6.2 ms
    // with at most 64 work-groups per layer; 128 or 256 are slower again
    static const int cap = [] { const char* c = getenv("DITREE_DENOISE_SPLITK_WGS"); return c ? atoi(c) : 64; }();
    while (sk < 16 && tiles * (sk * 2) <= cap && tiles * (sk * 2) <= SK_MAX_SLABS && nvc % (sk * 2) == 0 && nvc / (sk * 2) >= 2) sk *= 2;
    return sk;
  }
  void run_gemm(const ConvGemmParams& p_in, int fmt, hipStream_t s) {
    ConvGemmParams p = p_in;
    static const int strips = [] { const char* e = getenv("DITREE_XCD_STRIPS"); return e ? atoi(e) : 1; }();
    p.dbg = strips;                    // the halo / gemm16 kernels walk an XCD's tiles in strips of four tile columns (xcd_remap_strips)
    if (const int sk = pick_splitk(p, fmt); sk > 1) {
      if (sk_ws == nullptr) {
        sk_ws = (float*)dalloc((size_t)SK_MAX_SLABS * 65536 * sizeof(float), false);
        sk_cnt = (int*)dalloc(SK_MAX_SLABS * sizeof(int));
      }
      p.splitk = sk; p.sk_ws = sk_ws; p.sk_cnt = sk_cnt;
    }
    // shape contract of the tiles (split formats exist on the halo / gemm16 / small-Conv2d tiles only): an error, not an abort
    if (!conv_gemm_supported(p, fmt))
      throw std::runtime_error("denoiser layer of shape M " + std::to_string(p.M) + " x N " + std::to_string(p.N) + " x K " +
                               std::to_string(p.taps) + "*" + std::to_string(p.Cin) + " (L " + std::to_string(p.L) +
                               ") fits no tile of this precision");
    if (!prof_on) { launch_conv_gemm(p, fmt, s); return; }
    if (prof_used + 2 > prof_ev.size()) {
      const size_t old = prof_ev.size();
      prof_ev.resize(old + 4096);
      for (size_t i = old; i < prof_ev.size(); ++i) hipEventCreate(&prof_ev[i]);
    }
    const double fl = 2.0 * (double)p.M * (double)p.N * (double)p.taps * (double)p.Cin;
    if (prof_mode == 2) {
      // dominant kernel only: one (start, end) pair around each run of back-to-back halo launches, so the timed
      // region carries ~20 marker packets per denoiser call instead of ~120
      const int dom = fmt_st(ufmt) == ST_F32 ? 1 : 0;        // f32 instantiation: everything runs on conv_gemm_kernel
      if (conv_gemm_kind(p, fmt) != dom || fmt != ufmt) { close_run(); launch_conv_gemm(p, fmt, s); return; }
      if (run_open && prof_last_stream == s) {
        launch_conv_gemm(p, fmt, s);
        prof_pairs.back().flops += fl;
        prof_pairs.back().launches += 1;
        return;
      }
      close_run();
      const size_t st = prof_used++;
      run_end = prof_used++;
      hipEventRecord(prof_ev[st], s);
      launch_conv_gemm(p, fmt, s);
      prof_pairs.push_back(ProfRec{st, run_end, dom, fl, 1});
      run_open = true;
      prof_last_stream = s;
      return;
    }
    size_t start;
    if (prof_chain && prof_last_stream == s && prof_used > 0) {
      start = prof_used - 1;
    } else {
      start = prof_used++;
      hipEventRecord(prof_ev[start], s);
    }
    launch_conv_gemm(p, fmt, s);
    const size_t end = prof_used++;
    hipEventRecord(prof_ev[end], s);
    prof_pairs.push_back(ProfRec{start, end, conv_gemm_kind(p, fmt), fl, 1});
    prof_chain = true;
    prof_last_stream = s;
  }
  void prof_collect() {
    close_run();
    if (prof_used == 0) return;
    hipDeviceSynchronize();                                  // events live on several streams
    for (auto& pr : prof_pairs) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, prof_ev[pr.a], prof_ev[pr.b]) == hipSuccess) prof_ms_done[pr.kind] += ms;
      prof_launches_done[pr.kind] += pr.launches;
      prof_flops_done[pr.kind] += pr.flops;
    }
    prof_pairs.clear();
    prof_used = 0;
    prof_chain = false;
  }
  int es() const { return fmt_es(ufmt); }                   // element bytes of a U-Net activation plane
  int ees() const { return fmt_es(efmt); }                  // ... of an encoder activation
  int planes() const { return fmt_split(ufmt) ? 2 : 1; }

  const HostParam& P_(const std::string& name) const {
    auto it = params.find(name);
    if (it == params.end()) throw std::runtime_error("missing parameter " + name);
    return it->second;
  }
  bool has(const std::string& name) const { return params.count(name) != 0; }

  void* dalloc(size_t bytes, bool zero = true) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMalloc failed");
    if (zero && hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMemset failed");
    allocs.push_back(p);
    return p;
  }
  void free_workspace() {
    for (void* p : allocs) hipFree(p);
    allocs.clear();
    sk_ws = nullptr;                   // (allocated through dalloc: freed with the rest)
    sk_cnt = nullptr;
    named.clear();
    dev_w.clear();
    dev_f.clear();
    enc_ops.clear();
    unet_ops.clear();
    film_op = nullptr;
    sat_flags = sat_cond = sat_sample = nullptr;
    sat_names.clear();
    prec = -1;
    Bmax = 0;
    Buser = 0;
    temb_t = -1.0f;
  }

  float* upload_f32(const std::string& key, const float* src, int64_t n) {
    auto it = dev_f.find(key);
    if (it != dev_f.end()) return it->second;
    float* d = (float*)dalloc((size_t)n * 4, false);
    if (hipMemcpy(d, src, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("memcpy");
    dev_f[key] = d;
    return d;
  }
  float* vec(const std::string& name) {
    const HostParam& hp = P_(name);
    return upload_f32(name, hp.data, hp.n);
  }

  // Pack a GEMM weight [Npad][T*Cin_pad] in format `fmt` from get(n, t, ci): f32, or 16-bit elements -- for a split
  // format two planes hi = rnd16(w) and lo = rnd16(w - hi).  f16 has a narrow range: the matrix is stored multiplied by a
  // power of two that puts its largest magnitude in [2^13, 2^14); the kernels multiply the accumulator by 1 / that.
  template <class F>
  Packed pack(const std::string& key, int fmt, int N, int T, int Cin_pad, F get) {
    auto it = dev_w.find(key);
    if (it != dev_w.end()) return it->second;
    const int Npad = (N + 255) / 256 * 256;
    const size_t K = (size_t)T * Cin_pad, total = (size_t)Npad * K;
    const int st = fmt_st(fmt), e = fmt_es(fmt), npl = fmt_split(fmt) ? 2 : 1;
    std::vector<uint8_t> host(total * e * npl, 0);
    const int nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    float mul = 1.0f;
    if (st == ST_F16) {
      std::vector<float> mx(nthreads, 0.0f);
      std::vector<std::thread> th;
      for (int ti = 0; ti < nthreads; ++ti)
        th.emplace_back([&, ti]() {
          float m = 0.f;
          for (int n = ti; n < N; n += nthreads)
            for (int t = 0; t < T; ++t)
              for (int ci = 0; ci < Cin_pad; ++ci) m = std::max(m, std::fabs(get(n, t, ci)));
          mx[ti] = m;
        });
      for (auto& t : th) t.join();
      const float m = *std::max_element(mx.begin(), mx.end());
      if (m > 0.f && std::isfinite(m)) {
        int ex;
        std::frexp(m, &ex);                      // m = f * 2^ex, f in [0.5, 1)
        mul = std::ldexp(1.0f, 14 - ex);         // m * mul in [2^13, 2^14)
      }
    }
    std::vector<std::thread> th;
    for (int ti = 0; ti < nthreads; ++ti) {
      th.emplace_back([&, ti]() {
        for (int n = ti; n < N; n += nthreads) {
          for (int t = 0; t < T; ++t)
            for (int ci = 0; ci < Cin_pad; ++ci) {
              const float v = get(n, t, ci) * mul;
              const size_t idx = (size_t)n * K + (size_t)t * Cin_pad + ci;
              if (st == ST_F32) { ((float*)host.data())[idx] = v; continue; }
              uint16_t* hi = (uint16_t*)host.data();
              uint16_t* lo = hi + total;
              if (st == ST_BF16) {
                hi[idx] = f2bf_host(v);
                if (npl == 2) lo[idx] = f2bf_host(v - bf2f_host(hi[idx]));
              } else {
                hi[idx] = f2h_host(v);
                if (npl == 2) lo[idx] = f2h_host(v - h2f_host(hi[idx]));
              }
            }
        }
      });
    }
    for (auto& t : th) t.join();
    Packed pk;
    pk.p = dalloc(total * e * npl, false);
    pk.plane = npl == 2 ? (long long)(total * e) : 0;
    pk.scale = 1.0f / mul;
    if (hipMemcpy(pk.p, host.data(), total * e * npl, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("memcpy");
    dev_w[key] = pk;
    return pk;
  }
  static void set_w(ConvGemmParams& p, const Packed& pk) { p.W = pk.p; p.w_plane = pk.plane; p.w_scale = pk.scale; }

  Act make_act(const std::string& name, int L, int C, bool padded = true) {
    Act a;
    a.L = L; a.C = C; a.ld = C; a.coff = 0; a.padded = padded; a.fmt = ufmt;
    const size_t bytes = (size_t)Bmax * a.Lp() * C * es();
    a.plane = planes() == 2 ? (long long)bytes : 0;
    // split formats reach the lo plane through a 32-bit buffer offset from the tile base (2 GiB descriptor range)
    if (planes() == 2 && bytes >= 0x7f000000ull) throw std::runtime_error("activation plane of 2 GiB or more: lower the reserved batch");
    a.p = dalloc(bytes * planes());
    named[name] = a;
    return a;
  }
  static void set_in(ConvGemmParams& p, const Act& in) { p.a_plane = in.plane; }
  static void set_out(ConvGemmParams& p, const Act& out) { p.out_plane = out.plane; }
  static Act view(const Act& base, int coff, int C, const char* = nullptr) {
    Act v = base;
    v.coff = coff;
    v.C = C;
    return v;
  }
  const char* aptr(const Act& a) const { return (const char*)a.p + (size_t)a.coff * es(); }

  // ------------------------------------------------------------------ GEMM op builders
  // Conv1d(k = 3, pad 1) [+ GroupNorm(8) + Mish (+ FiLM | + residual)] on padded activations.
  void add_conv3(std::vector<std::function<void(int, int, hipStream_t)>>& ops, const std::string& wname, const Act& in,
                 const Act& out, int mode, const std::string& gn, int film_off, const Act* res) {
    const HostParam& w = P_(wname + ".weight");
    const int Cout = (int)w.dims[0], Cin = (int)w.dims[1];
    const float* wd = w.data;
    const Packed wp = pack(wname, ufmt, Cout, 3, Cin, [=](int n, int t, int ci) { return wd[((size_t)n * Cin + ci) * 3 + t]; });
    ConvGemmParams p{};
    p.A = aptr(in); p.lda = in.ld; p.in_Lp = in.Lp(); p.in_stride = 1; p.in_off = 0; p.taps = 3; p.Cin = Cin;
    set_w(p, wp); set_in(p, in); set_out(p, out);
    p.sat = sat_slot(wname, ufmt);
    p.Out = (void*)aptr(out); p.ldc = out.ld; p.out_Lp = out.Lp(); p.out_stride = 1; p.out_off = 1; p.out_coff = 0;
    p.L = in.L; p.N = Cout; p.bias = vec(wname + ".bias"); p.mode = mode; p.eps = 1e-5f;
    if (mode >= MODE_GN_MISH) {
      p.gamma = vec(gn + ".weight"); p.beta = vec(gn + ".bias"); p.group_ch = Cout / 8;
    }
    if (mode == MODE_GN_MISH_FILM) { p.film = film; p.film_ld = film_cols; p.film_off = film_off; }
    if (mode == MODE_GN_MISH_RES) {
      p.Res = aptr(*res); p.ldres = res->ld; p.res_Lp = res->Lp(); p.res_off = res->padded ? 1 : 0; p.res_plane = res->plane;
    }
    emit_block(ops, p, out, in.L);
  }
  // A GEMM whose epilogue is GroupNorm(8) + Mish (+ FiLM | + residual).  The fused epilogue needs whole GroupNorm
  // groups inside a 256-channel tile: 64, 128 or 256 channels per group (C_out 512 / 1024 / 2048, the `large`
  // denoiser).  Other sizes run conv + bias in the GEMM and the normalisation / Mish / FiLM / residual in gn1d_kernel.
  void emit_block(std::vector<std::function<void(int, int, hipStream_t)>>& ops, const ConvGemmParams& p, const Act& out, int L) {
    const int Cout = p.N, mode = p.mode, uf = ufmt;
    const int gch = Cout / 8;
    // ... and a lane's four accumulator rows inside one sample: 16 | L, or L = 8 / 4 on the 16-bit tiles (the ant config's lower
    // levels; DITREE_GN_SHORT_UNFUSED=1 keeps the round-3 two-launch form for A/B runs)
    static const bool short_unfused = [] { const char* e = getenv("DITREE_GN_SHORT_UNFUSED"); return e && atoi(e) != 0; }();
    const bool short_fusable = (L == 8 || L == 4) && fmt_st(uf) != ST_F32 && !short_unfused;
    const bool fused = mode < MODE_GN_MISH ||
                       ((Cout & 255) == 0 && (gch == 64 || gch == 128 || gch == 256) && ((L & 15) == 0 || short_fusable));
    // short levels fuse on the gemm16 tile only: whole 256-row tiles (always so for the split formats, batch by batch otherwise)
    const bool fused_needs_tiles = fused && mode >= MODE_GN_MISH && (L & 15) != 0;
    ConvGemmParams qf = p, q = p;
    q.mode = MODE_BIAS;
    const float* film_p = p.film;
    const int film_ld_ = p.film_ld, film_off_ = p.film_off, ldres = p.ldres, res_Lp = p.res_Lp, res_off = p.res_off;
    const void* resp = p.Res;
    const float *ga = p.gamma, *be = p.beta;
    void* xo = (void*)out.p;
    const int ld = out.ld, oLp = out.Lp(), ocoff = out.coff;
    const long long xpl = out.plane, rpl = p.res_plane;
    int* gsat = p.sat;                 // the normalisation pass reports under the layer's name as well
    ops.push_back([=, this](int, int Bp, hipStream_t s) mutable {
      if (fused && (!fused_needs_tiles || ((Bp * L) & 255) == 0)) {
        qf.M = Bp * L;
        run_gemm(qf, uf, s);
        return;
      }
      q.M = Bp * L;
      run_gemm(q, uf, s);
      note_other();
      launch_gn1d(xo, ld, oLp, 1, ocoff, L, Cout, ga, be, 1e-5f, mode, film_p, film_ld_, film_off_, resp, ldres, res_Lp,
                  res_off, Bp, uf, xpl, rpl, s, gsat);
    });
  }
  // Conv1d(k = 1) residual projection.
  void add_conv1(std::vector<std::function<void(int, int, hipStream_t)>>& ops, const std::string& wname, const Act& in,
                 const Act& out) {
    const HostParam& w = P_(wname + ".weight");
    const int Cout = (int)w.dims[0], Cin = (int)w.dims[1];
    const float* wd = w.data;
    const Packed wp = pack(wname, ufmt, Cout, 1, Cin, [=](int n, int, int ci) { return wd[(size_t)n * Cin + ci]; });
    ConvGemmParams p{};
    p.A = aptr(in); p.lda = in.ld; p.in_Lp = in.Lp(); p.in_stride = 1; p.in_off = 1; p.taps = 1; p.Cin = Cin;
    set_w(p, wp); set_in(p, in); set_out(p, out);
    p.sat = sat_slot(wname, ufmt);
    p.Out = (void*)aptr(out); p.ldc = out.ld; p.out_Lp = out.Lp(); p.out_stride = 1; p.out_off = 1;
    p.L = in.L; p.N = Cout; p.bias = vec(wname + ".bias"); p.mode = MODE_BIAS;
    const int L = in.L, uf = ufmt;
    ops.push_back([this, p, L, uf](int, int Bp, hipStream_t s) mutable {
      p.M = Bp * L;
      run_gemm(p, uf, s);
    });
  }
  // Downsample1d: Conv1d(C, C, 3, stride 2, pad 1)  (conv1d_components.py:7-13)
  void add_down(std::vector<std::function<void(int, int, hipStream_t)>>& ops, const std::string& wname, const Act& in,
                const Act& out) {
    const HostParam& w = P_(wname + ".weight");
    const int Cout = (int)w.dims[0], Cin = (int)w.dims[1];
    const float* wd = w.data;
    const Packed wp = pack(wname, ufmt, Cout, 3, Cin, [=](int n, int t, int ci) { return wd[((size_t)n * Cin + ci) * 3 + t]; });
    ConvGemmParams p{};
    p.A = aptr(in); p.lda = in.ld; p.in_Lp = in.Lp(); p.in_stride = 2; p.in_off = 0; p.taps = 3; p.Cin = Cin;
    set_w(p, wp); set_in(p, in); set_out(p, out);
    p.sat = sat_slot(wname, ufmt);
    p.Out = (void*)aptr(out); p.ldc = out.ld; p.out_Lp = out.Lp(); p.out_stride = 1; p.out_off = 1;
    p.L = out.L; p.N = Cout; p.bias = vec(wname + ".bias"); p.mode = MODE_BIAS;
    const int L = out.L, uf = ufmt;
    ops.push_back([this, p, L, uf](int, int Bp, hipStream_t s) mutable {
      p.M = Bp * L;
      run_gemm(p, uf, s);
    });
  }
  // Upsample1d: ConvTranspose1d(C, C, 4, 2, 1) as two 2-tap GEMMs (even / odd outputs)
  // out[2m] = W1^T x[m] + W3^T x[m-1];  out[2m+1] = W0^T x[m+1] + W2^T x[m]   (conv1d_components.py:15-21)
  void add_up(std::vector<std::function<void(int, int, hipStream_t)>>& ops, const std::string& wname, const Act& in,
              const Act& out) {
    const HostParam& w = P_(wname + ".weight");                  // [Cin][Cout][4]
    const int Cin = (int)w.dims[0], Cout = (int)w.dims[1];
    const float* wd = w.data;
    for (int par = 0; par < 2; ++par) {
      const int k0 = par == 0 ? 3 : 2, k1 = par == 0 ? 1 : 0;     // tap 0 -> earlier input row
      const Packed wp = pack(wname + (par ? ".odd" : ".even"), ufmt, Cout, 2, Cin, [=](int n, int t, int ci) {
        return wd[((size_t)ci * Cout + n) * 4 + (t == 0 ? k0 : k1)];
      });
      ConvGemmParams p{};
      p.A = aptr(in); p.lda = in.ld; p.in_Lp = in.Lp(); p.in_stride = 1; p.in_off = par; p.taps = 2; p.Cin = Cin;
      set_w(p, wp); set_in(p, in); set_out(p, out);
      p.sat = sat_slot(wname, ufmt);
      p.Out = (void*)aptr(out); p.ldc = out.ld; p.out_Lp = out.Lp(); p.out_stride = 2; p.out_off = 1 + par;
      p.L = in.L; p.N = Cout; p.bias = vec(wname + ".bias"); p.mode = MODE_BIAS;
      const int L = in.L, uf = ufmt;
      ops.push_back([this, p, L, uf](int, int Bp, hipStream_t s) mutable {
        p.M = Bp * L;
        run_gemm(p, uf, s);
      });
    }
  }

  // ConditionalResidualBlock1D (conditional_unet1d.py:41-142)
  void add_crb(const std::string& pre, const Act& in, const Act& out, int& film_cursor, const std::string& tag) {
    const int Cout = (int)P_(pre + ".blocks.0.block.0.weight").dims[0];
    Act h = make_act(tag + ".h", in.L, Cout);
    const int film_off = film_cursor;
    film_cursor += 2 * Cout;
    add_conv3(unet_ops, pre + ".blocks.0.block.0", in, h, MODE_GN_MISH_FILM, pre + ".blocks.0.block.1", film_off, nullptr);
    Act res = in;
    if (has(pre + ".residual_conv.weight")) {
      res = make_act(tag + ".res", in.L, Cout);
      add_conv1(unet_ops, pre + ".residual_conv", in, res);
    }
    add_conv3(unet_ops, pre + ".blocks.1.block.0", h, out, MODE_GN_MISH_RES, pre + ".blocks.1.block.1", 0, &res);
  }

  void build(int prec_, int Bmax_);
};

void DenoiserState::build(int prec_, int Bmax_) {
  free_workspace();
  prec = prec_;
  // DITREE_PREC_*  ->  formats of the U-Net and of the encoder (the split instantiations run the encoder split as well: its
  // stem is the fused kernel for both map sizes, 20 x 20 and 16 x 16).
  switch (prec) {
    case DITREE_PREC_BF16: ufmt = fmt_make(ST_BF16, false); efmt = ST_BF16; break;
    case DITREE_PREC_F32: ufmt = fmt_make(ST_F32, false); efmt = ST_F32; break;
    case DITREE_PREC_F16X3: ufmt = fmt_make(ST_F16, true); efmt = ufmt; break;
    case DITREE_PREC_BF16X3: ufmt = fmt_make(ST_BF16, true); efmt = ufmt; break;
    case DITREE_PREC_F16: ufmt = fmt_make(ST_F16, false); efmt = ST_F16; break;
    default: throw std::runtime_error("unknown precision");
  }
  sat_flags = (int*)dalloc(SAT_SLOTS * sizeof(int));
  sat_cond = sat_slot("cond_encoder input: Mish(time | map embedding | observation)", ufmt);
  sat_sample = sat_slot("sample (the noisy action sequence)", ufmt);
  const int C0 = dims[0], C1 = dims[1], C2 = dims[2];
  const int L0 = P, L1 = P / 2, L2 = P / 4;
  if (P % 16 != 0 || P < 16 || P > 256 || (256 % P) != 0) throw std::runtime_error("pred_horizon must be 16, 32, 64, 128 or 256");
  // rows of a batch are padded to whole work units: 16 samples (f32), or -- the 16-bit formats run every level on the 256-row
  // halo / gemm16 tiles (the split formats exist there only; the fused GroupNorm epilogue of the short levels needs whole
  // tiles, and a sample's result must not depend on the batch it is part of) -- as many as fill a tile at the shortest level
  // (P / 4 rows per sample: 64 samples at P = 16)
  bgran = fmt_st(ufmt) != ST_F32 ? std::max(16, 1024 / P) : 16;
  Buser = Bmax_;
  Bmax = (Bmax_ + bgran - 1) / bgran * bgran;
  {
    // The split kernels reach the lo plane through a 32-bit buffer offset, so an activation plane must stay below 2 GiB; and
    // nothing is gained by a workspace beyond a few thousand samples (every layer already launches >> 256 tiles).  Larger
    // calls run as sub-batches of `Bmax` samples (denoise_core): the reserved batch is a capacity, not a workspace size.
    const long long per_sample = std::max({(long long)(L2 + 2) * 2 * C2, (long long)(L1 + 2) * 2 * C1, (long long)(L0 + 2) * C0,
                                           (long long)L0 * 64}) * fmt_es(ufmt);
    long long cap = std::min<long long>(8192, 0x7f000000LL / per_sample);
    cap = cap / 256 * 256;
    if (cap < 256) throw std::runtime_error("denoiser too large for one 256-sample work unit");
    if (Bmax > cap) Bmax = (int)cap;
  }
  if (fmt_split(ufmt) && ((C0 | C1 | C2) & 255) != 0)
    throw std::runtime_error("the split precisions need down_dims that are multiples of 256 (halo / gemm16 tiles)");
  x_cur = (float*)dalloc((size_t)Bmax * P * D * 4);
  temb = (float*)dalloc(256 * 4);
  condK = (cond_dim + 63) / 64 * 64;
  // the FiLM GEMM runs on whole 256-row tiles: rows padded (zero rows in, ignored rows out)
  const int Brows = (Bmax + 255) / 256 * 256;
  // split encoder: its fc layer runs on the gemm16 tiles too (rows and columns padded to 256)
  E_ld = fmt_split(efmt) ? (E + 255) / 256 * 256 : E;
  map_emb = (float*)dalloc((size_t)Brows * E_ld * 4);
  cond_plane = planes() == 2 ? (long long)Brows * condK * es() : 0;
  condA = dalloc((size_t)Brows * condK * es() * planes());

  // ---------------- FiLM: all cond_encoder Linear layers batched into one GEMM -------------------------
  std::vector<std::string> crbs = {"unet.down_modules.0.0", "unet.down_modules.0.1", "unet.down_modules.1.0",
                                   "unet.down_modules.1.1", "unet.down_modules.2.0", "unet.down_modules.2.1",
                                   "unet.mid_modules.0",    "unet.mid_modules.1",    "unet.up_modules.0.0",
                                   "unet.up_modules.0.1",   "unet.up_modules.1.0",   "unet.up_modules.1.1"};
  film_cols = 0;
  std::vector<int> film_offs;
  for (auto& c : crbs) {
    film_offs.push_back(film_cols);
    film_cols += (int)P_(c + ".cond_encoder.1.weight").dims[0];
  }
  film = (float*)dalloc((size_t)Brows * film_cols * 4);
  {
    std::vector<const float*> wsrc, bsrc;
    std::vector<int> rows;
    for (auto& c : crbs) {
      wsrc.push_back(P_(c + ".cond_encoder.1.weight").data);
      bsrc.push_back(P_(c + ".cond_encoder.1.bias").data);
      rows.push_back((int)P_(c + ".cond_encoder.1.weight").dims[0]);
    }
    std::vector<int> start(rows.size());
    for (size_t i = 0, a = 0; i < rows.size(); ++i) { start[i] = (int)a; a += rows[i]; }
    const int cd = cond_dim;
    const Packed wp = pack("film.all", ufmt, film_cols, 1, condK, [=](int n, int, int ci) {
      size_t i = std::upper_bound(start.begin(), start.end(), n) - start.begin() - 1;
      return ci < cd ? wsrc[i][(size_t)(n - start[i]) * cd + ci] : 0.0f;
    });
    std::vector<float> ball(film_cols);
    for (size_t i = 0; i < rows.size(); ++i) std::memcpy(ball.data() + start[i], bsrc[i], (size_t)rows[i] * 4);
    float* bd = upload_f32("film.bias", ball.data(), film_cols);
    ConvGemmParams p{};
    p.A = condA; p.lda = condK; p.in_Lp = 0; p.in_stride = 1; p.in_off = 0; p.taps = 1; p.Cin = condK;
    set_w(p, wp); p.a_plane = cond_plane;
    p.Out = film; p.ldc = film_cols; p.out_Lp = 0; p.out_stride = 1; p.out_off = 0;
    p.N = film_cols; p.bias = bd; p.mode = MODE_BIAS; p.out_f32 = 1;
    const int uf = ufmt;
    film_op = [this, p, uf](int, int Bp, hipStream_t s) mutable {
      const int Mr = fmt_st(uf) == ST_F32 ? Bp : (Bp + 255) / 256 * 256;
      p.M = Mr; p.L = Mr; p.in_Lp = Mr; p.out_Lp = Mr;
      run_gemm(p, uf, s);
    };
  }

  // ---------------- U-Net ----------------------------------------------------------------------------
  int fc = 0;   // film cursor follows `crbs` order
  // first layer input: im2col rows [x[l-1], x[l], x[l+1]] padded to 64 columns, unpadded rows
  Act a0;
  a0.L = L0; a0.C = 64; a0.ld = 64; a0.coff = 0; a0.padded = false; a0.fmt = ufmt;
  a0.plane = planes() == 2 ? (long long)Bmax * L0 * 64 * es() : 0;
  a0.p = dalloc((size_t)Bmax * L0 * 64 * es() * planes());
  named["a0"] = a0;
  {
    // down 0, block 1: Conv1d(D, C0, 3) as a 1-tap GEMM over the im2col rows
    const std::string pre = "unet.down_modules.0.0";
    const HostParam& w = P_(pre + ".blocks.0.block.0.weight");
    const float* wd = w.data;
    const int Dd = D;
    const Packed wp = pack(pre + ".blocks.0.block.0", ufmt, C0, 1, 64, [=](int n, int, int ci) {
      if (ci >= 3 * Dd) return 0.0f;
      const int t = ci / Dd, d = ci - t * Dd;
      return wd[((size_t)n * Dd + d) * 3 + t];
    });
    Act h = make_act("d0b1.h", L0, C0);
    ConvGemmParams p{};
    p.A = a0.p; p.lda = 64; p.in_Lp = L0; p.in_stride = 1; p.in_off = 0; p.taps = 1; p.Cin = 64;
    set_w(p, wp); set_in(p, a0); set_out(p, h);
    p.sat = sat_slot(pre + ".blocks.0.block.0", ufmt);
    p.Out = h.p; p.ldc = h.ld; p.out_Lp = h.Lp(); p.out_stride = 1; p.out_off = 1;
    p.L = L0; p.N = C0; p.bias = vec(pre + ".blocks.0.block.0.bias"); p.mode = MODE_GN_MISH_FILM; p.eps = 1e-5f;
    p.gamma = vec(pre + ".blocks.0.block.1.weight"); p.beta = vec(pre + ".blocks.0.block.1.bias"); p.group_ch = C0 / 8;
    p.film = film; p.film_ld = film_cols; p.film_off = film_offs[0];
    const int L = L0;
    emit_block(unet_ops, p, h, L0);
    // residual Conv1d(D, C0, 1): centre-tap columns of the same rows
    const HostParam& wr = P_(pre + ".residual_conv.weight");
    const float* wrd = wr.data;
    const Packed wrp = pack(pre + ".residual_conv", ufmt, C0, 1, 64, [=](int n, int, int ci) {
      return (ci >= Dd && ci < 2 * Dd) ? wrd[(size_t)n * Dd + (ci - Dd)] : 0.0f;
    });
    Act res = make_act("d0b1.res", L0, C0);
    ConvGemmParams q{};
    q.A = a0.p; q.lda = 64; q.in_Lp = L0; q.in_stride = 1; q.in_off = 0; q.taps = 1; q.Cin = 64;
    set_w(q, wrp); set_in(q, a0); set_out(q, res);
    q.sat = sat_slot(pre + ".residual_conv", ufmt);
    q.Out = res.p; q.ldc = res.ld; q.out_Lp = res.Lp(); q.out_stride = 1; q.out_off = 1;
    q.L = L0; q.N = C0; q.bias = vec(pre + ".residual_conv.bias"); q.mode = MODE_BIAS;
    const int uf = ufmt;
    unet_ops.push_back([this, q, L, uf](int, int Bp, hipStream_t s) mutable { q.M = Bp * L; run_gemm(q, uf, s); });
    Act o = make_act("d0b1.out", L0, C0);
    add_conv3(unet_ops, pre + ".blocks.1.block.0", h, o, MODE_GN_MISH_RES, pre + ".blocks.1.block.1", 0, &res);
    fc = film_offs[1];
  }
  Act cat1 = make_act("cat1", L1, 2 * C1);        // [x_up (C1) | skip1 (C1)] at L1
  Act cat0 = make_act("cat0", L2, 2 * C2);        // [x_mid (C2) | skip2 (C2)] at L2
  Act d0o = named["d0b1.out"];
  Act skip0 = make_act("skip0", L0, C0);
  add_crb("unet.down_modules.0.1", d0o, skip0, fc, "d0b2");
  Act d1in = make_act("d1.in", L1, C0);
  add_down(unet_ops, "unet.down_modules.0.2.conv", skip0, d1in);
  Act d1o = make_act("d1b1.out", L1, C1);
  add_crb("unet.down_modules.1.0", d1in, d1o, fc, "d1b1");
  Act skip1 = view(cat1, C1, C1);
  named["skip1"] = skip1;
  add_crb("unet.down_modules.1.1", d1o, skip1, fc, "d1b2");
  Act d2in = make_act("d2.in", L2, C1);
  add_down(unet_ops, "unet.down_modules.1.2.conv", skip1, d2in);
  Act d2o = make_act("d2b1.out", L2, C2);
  add_crb("unet.down_modules.2.0", d2in, d2o, fc, "d2b1");
  Act skip2 = view(cat0, C2, C2);
  named["skip2"] = skip2;
  add_crb("unet.down_modules.2.1", d2o, skip2, fc, "d2b2");
  Act m1 = make_act("mid1.out", L2, C2);
  add_crb("unet.mid_modules.0", skip2, m1, fc, "mid1");
  Act m2 = view(cat0, 0, C2);
  named["mid2.out"] = m2;
  add_crb("unet.mid_modules.1", m1, m2, fc, "mid2");
  Act u0a = make_act("u0b1.out", L2, C1);
  add_crb("unet.up_modules.0.0", cat0, u0a, fc, "u0b1");
  Act u0b = make_act("u0b2.out", L2, C1);
  add_crb("unet.up_modules.0.1", u0a, u0b, fc, "u0b2");
  Act up0 = view(cat1, 0, C1);
  named["up0.out"] = up0;
  add_up(unet_ops, "unet.up_modules.0.2.conv", u0b, up0);
  Act u1a = make_act("u1b1.out", L1, C0);
  add_crb("unet.up_modules.1.0", cat1, u1a, fc, "u1b1");
  Act u1b = make_act("u1b2.out", L1, C0);
  add_crb("unet.up_modules.1.1", u1a, u1b, fc, "u1b2");
  Act fin = make_act("final.in", L0, C0);
  add_up(unet_ops, "unet.up_modules.1.2.conv", u1b, fin);
  final_h = make_act("final.h", L0, C0);
  add_conv3(unet_ops, "unet.final_conv.0.block.0", fin, final_h, MODE_GN_MISH, "unet.final_conv.0.block.1", 0, nullptr);
  vec("unet.final_conv.1.weight");
  vec("unet.final_conv.1.bias");
  vec("unet.diffusion_step_encoder.1.weight");
  vec("unet.diffusion_step_encoder.1.bias");
  vec("unet.diffusion_step_encoder.3.weight");
  vec("unet.diffusion_step_encoder.3.bias");

  // ---------------- encoder: ResNet-18 with GroupNorm(C/16), NHWC, im2col + GEMM + GN kernels ------------
  // Every op works on a sub-batch [b0, b0 + Bn) with its own im2col / GEMM-output scratch region, so that
  // several sub-batches can run concurrently on different streams (the layers are small: 8..100 tiles
  // each, far fewer than the 256 CUs).
  {
    const std::string R = "encoder.resnet18.";
    const int pr = efmt;
    const size_t E_ = ees();
    zero_row = dalloc(256);
    sub_cap = (Bmax + ENC_SUBS - 1) / ENC_SUBS;
    sub_cap = (sub_cap + 15) / 16 * 16;
    const size_t col_per_sample = 25 * 576;                         // largest im2col footprint per sample (layer1)
    const size_t gout_per_sample = 9216;                            // f32 GEMM output per sample: stem 100*64, or up to 8 split-K slabs of 9*128
    col_region = (size_t)sub_cap * col_per_sample;
    gout_region = (size_t)sub_cap * gout_per_sample;
    char* col = (char*)dalloc(col_region * ENC_SUBS * E_);
    float* gout = (float*)dalloc(gout_region * ENC_SUBS * 4);
    const size_t colreg = col_region, goutreg = gout_region;
    auto ebuf = [&](const std::string& name, int HW, int C) {
      Act a;
      a.L = HW; a.C = C; a.ld = C; a.coff = 0; a.padded = false; a.fmt = efmt;
      const size_t rows = HW == 1 ? (size_t)Brows : (size_t)Bmax;   // the pooled vector feeds a GEMM on whole 256-row tiles
      const size_t bytes = rows * HW * C * ees();
      a.plane = fmt_split(efmt) ? (long long)bytes : 0;
      a.p = dalloc(bytes * (fmt_split(efmt) ? 2 : 1));
      named[name] = a;
      return a;
    };
    auto live_taps = [](int k, int stride, int pad, int H, int OH) {
      TapList tl{};
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
          bool lh = false, lw = false;
          for (int o = 0; o < OH; ++o) {
            const int ih = o * stride + kh - pad, iw = o * stride + kw - pad;
            lh |= (ih >= 0 && ih < H);
            lw |= (iw >= 0 && iw < H);
          }
          if (lh && lw) { tl.kh[tl.n] = (signed char)kh; tl.kw[tl.n] = (signed char)kw; ++tl.n; }
        }
      return tl;
    };
    // conv (+ optional folding of the 3 identical input channels) -> f32 GEMM output in the region's `gout`.
    // in == nullptr: the source is the caller's f32 local map (lm_ptr), C = 1.
    auto conv2d = [&](const std::string& wname, const void* in, long long in_plane, int H, int Cin, int Cout, int k, int stride,
                      int pad, int OH, bool fold_in) {
      const HostParam& w = P_(wname + ".weight");
      const float* wd = w.data;
      const int Cw = (int)w.dims[1];
      const TapList tl = live_taps(k, stride, pad, H, OH);
      const int K = tl.n * Cin, Kpad = (K + 63) / 64 * 64;
      if ((size_t)OH * OH * Kpad > col_per_sample || (size_t)OH * OH * Cout > gout_per_sample)
        throw std::runtime_error("encoder scratch region too small for " + wname);
      const Packed wp = pack(wname, efmt, Cout, 1, Kpad, [=](int n, int, int kk) {
        if (kk >= K) return 0.0f;
        const int c = kk % Cin, t = kk / Cin, kw = tl.kw[t], kh = tl.kh[t];
        if (!fold_in) return wd[(((size_t)n * Cw + c) * k + kh) * k + kw];
        float s = 0.f;                                       // x.repeat(1,3,1,1): identical channels fold into one
        for (int cc = 0; cc < Cw; ++cc) s += wd[(((size_t)n * Cw + cc) * k + kh) * k + kw];
        return s;
      });
      const float** lm_slot = &lm_ptr;
      const bool implicit = (in != nullptr) && (Cin % 64 == 0) && tl.n <= 12;
      if (!implicit && fmt_split(efmt)) throw std::runtime_error("split encoder: " + wname + " needs the im2col path");
      const void* zr = zero_row;
      enc_ops.push_back([=, this](int b0, int Bn, int reg, hipStream_t s) {
        char* colr = col + (size_t)reg * colreg * E_;
        float* goutr = gout + (size_t)reg * goutreg;
        ConvGemmParams p{};
        const int M = Bn * OH * OH;
        if (implicit) {
          // implicit GEMM: the kernel gathers the (tap, channel-chunk) rows itself, no im2col pass
          p.A = (const char*)in + (size_t)b0 * H * H * Cin * E_; p.lda = Cin; p.taps = tl.n; p.Cin = Cin; p.a_plane = in_plane;
          p.c2d = 1; p.c2_H = H; p.c2_W = H; p.c2_OW = OH; p.c2_OHW = OH * OH; p.c2_stride = stride; p.c2_pad = pad;
          for (int t = 0; t < tl.n; ++t) { p.c2_kh[t] = tl.kh[t]; p.c2_kw[t] = tl.kw[t]; }
          p.zero = zr;
          // split-K: these layers have few tiles and long K loops; spread the K-steps over idle CUs.  The
          // factor depends on the layer only (sized for 1024 candidates), never on the batch, so that a
          // candidate's result does not depend on which other candidates share its launch (bitwise).
          const int nk_total = tl.n * (Cin / 64);
          const int tiles = ((1024 * OH * OH + 255) / 256) * ((Cout + 255) / 256);
          static int sk_target = -1, sk_cap = 8;          // A/B knobs: DITREE_SPLITK_TARGET (work-groups aimed at), DITREE_SPLITK_CAP
          if (sk_target < 0) {
            const char* e = getenv("DITREE_SPLITK_TARGET");
            sk_target = (e && atoi(e) > 0) ? atoi(e) : 256;      // one work-group per CU
            const char* c = getenv("DITREE_SPLITK_CAP");
            if (c && atoi(c) > 0) sk_cap = atoi(c);
          }
          int sk = std::max(1, std::min(std::min(sk_cap, nk_total / 2), std::max(1, sk_target / tiles)));
          while (sk > 1 && (size_t)sk * OH * OH * Cout > gout_per_sample) --sk;
          if (sk > 1) {                                   // every split must own at least one K-step
            const int per = (nk_total + sk - 1) / sk;
            sk = (nk_total + per - 1) / per;
          }
          // measured on MI355X at B = 1024: 28.2 ms per round with split-K vs 29.6 without; DITREE_SPLITK=0 disables
          static int use_split = -1;
          if (use_split < 0) { const char* e = getenv("DITREE_SPLITK"); use_split = (e && !atoi(e)) ? 0 : 1; }
          if (conv2d_small_eligible(pr)) {
            // 64 x 64 tiles, three work-groups per CU: split only the layers that cannot fill those slots
            static int small_target = -1;
            if (small_target < 0) { const char* e = getenv("DITREE_C2D_TARGET"); small_target = (e && atoi(e) > 0) ? atoi(e) : 768; }
            const int tiles64 = ((1024 * OH * OH + 63) / 64) * (Cout / 64);
            sk = std::max(1, std::min(std::min(8, nk_total / 2), small_target / tiles64));
            while (sk > 1 && (size_t)sk * OH * OH * Cout > gout_per_sample) --sk;
            if (sk > 1) {
              const int per = (nk_total + sk - 1) / sk;
              sk = (nk_total + per - 1) / per;
            }
          }
          if (!use_split) sk = 1;
          p.splitk = sk;
          p.slab_stride = (long long)M * Cout;
          last_splitk[reg] = sk;
          last_slab[reg] = p.slab_stride;
        } else {
          last_splitk[reg] = 1;
          last_slab[reg] = 0;
        }
        if (!implicit) {
          note_other();
          if (in == nullptr)
            launch_im2col2d(*lm_slot + (size_t)b0 * H * H, true, colr, Bn, H, H, 1, tl, stride, pad, OH, OH, Kpad, pr, s);
          else
            launch_im2col2d((const char*)in + (size_t)b0 * H * H * Cin * E_, false, colr, Bn, H, H, Cin, tl, stride, pad,
                            OH, OH, Kpad, pr, s);
          p.A = colr; p.lda = Kpad; p.taps = 1; p.Cin = Kpad;
        }
        p.in_Lp = M; p.in_stride = 1; p.in_off = 0;
        set_w(p, wp);
        p.Out = goutr; p.ldc = Cout; p.out_Lp = M; p.out_stride = 1; p.out_off = 0;
        p.L = M; p.M = M; p.N = Cout; p.mode = MODE_BIAS; p.out_f32 = 1;
        run_gemm(p, pr, s);
      });
    };
    auto gn = [&](const std::string& gname, const Act& out, const Act* res, bool relu) {
      float* ga = vec(gname + ".weight");
      float* be = vec(gname + ".bias");
      const char* rp = res ? (const char*)res->p : nullptr;
      char* op = (char*)out.p;
      const int HW = out.L, C = out.C;
      const long long rpl = res ? res->plane : 0, opl = out.plane;
      {                                                  // shape contract of gn2d_kernel, checked when the plan is built
        const int pp = (C >= 64 && C <= 1024 && (C & (C - 1)) == 0) ? 256 / (C >> 2) : 0;
        if (pp == 0 || (HW + pp - 1) / pp > 7)
          throw std::runtime_error("encoder GroupNorm: unsupported map (" + std::to_string(HW) + " pixels x " + std::to_string(C) + " channels)");
      }
      int* gsat = sat_slot(gname, efmt);
      enc_ops.push_back([=, this](int b0, int Bn, int reg, hipStream_t s) {
        note_other();
        const size_t off = (size_t)b0 * HW * C * E_;
        launch_gn2d(gout + (size_t)reg * goutreg, last_splitk[reg], last_slab[reg], ga, be, rp ? rp + off : nullptr,
                    relu ? 1 : 0, op + off, Bn, HW, C, 1e-5f, pr, rpl, opl, s, gsat);
      });
    };
    const int H0 = lm;
    auto osz = [](int h, int k, int s, int p) { return (h + 2 * p - k) / s + 1; };
    const int H1 = osz(H0, 7, 2, 3);                  // 10
    const int H2 = osz(H1, 3, 2, 1);                  // 5
    Act c1 = ebuf("enc.c1", H1 * H1, 64);
    Act pool = ebuf("enc.pool", H2 * H2, 64);
    if (H0 == 20 || H0 == 16) {
      // one launch: conv 7x7/2 (input channels folded) + GroupNorm + ReLU + max-pool
      const HostParam& w = P_(R + "conv1.weight");
      const int Cw = (int)w.dims[1];
      std::vector<float> wf(64 * 49);
      for (int n = 0; n < 64; ++n)
        for (int t = 0; t < 49; ++t) {
          float sacc = 0.f;
          for (int cc = 0; cc < Cw; ++cc) sacc += w.data[((size_t)n * Cw + cc) * 49 + t];
          wf[t * 64 + n] = sacc;                       // tap-major: lanes (channels) read consecutive floats
        }
      float* wdev = upload_f32(R + "conv1.weight#folded", wf.data(), 64 * 49);
      float* ga = vec(R + "bn1.weight");
      float* be = vec(R + "bn1.bias");
      char* op = (char*)pool.p;
      const float** lm_slot = &lm_ptr;
      const long long ppl = pool.plane;
      int* ssat = sat_slot(R + "conv1", efmt);
      enc_ops.push_back([=, this](int b0, int Bn, int, hipStream_t s) {
        note_other();
        launch_encoder_stem(*lm_slot + (size_t)b0 * H0 * H0, H0, wdev, ga, be, op + (size_t)b0 * H2 * H2 * 64 * E_, Bn, 1e-5f, pr, ppl, s, ssat);
      });
    } else {
    conv2d(R + "conv1", nullptr, 0, H0, 1, 64, 7, 2, 3, H1, true);
    gn(R + "bn1", c1, nullptr, true);
    {
      const char* ip = (const char*)c1.p; char* op = (char*)pool.p;
      const int a = H1, b2 = H2;
      enc_ops.push_back([=, this](int b0, int Bn, int, hipStream_t s) {
        note_other();
        launch_maxpool2d(ip + (size_t)b0 * a * a * 64 * E_, op + (size_t)b0 * b2 * b2 * 64 * E_, Bn, a, a, 64, b2, b2, pr, s);
      });
    }
    }
    Act cur = pool;
    int Hc = H2, Cc = 64;
    const int chans[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li) {
      for (int bi = 0; bi < 2; ++bi) {
        const std::string pre = R + "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
        const int Cout = chans[li];
        const int stride = (bi == 0 && li > 0) ? 2 : 1;
        const int Ho = osz(Hc, 3, stride, 1);
        const std::string tag = "enc.l" + std::to_string(li + 1) + "." + std::to_string(bi);
        Act t1 = ebuf(tag + ".t", Ho * Ho, Cout);
        conv2d(pre + ".conv1", cur.p, cur.plane, Hc, Cc, Cout, 3, stride, 1, Ho, false);
        gn(pre + ".bn1", t1, nullptr, true);
        Act idt = cur;
        if (has(pre + ".downsample.0.weight")) {
          Act ds = ebuf(tag + ".ds", Ho * Ho, Cout);
          conv2d(pre + ".downsample.0", cur.p, cur.plane, Hc, Cc, Cout, 1, stride, 0, Ho, false);
          gn(pre + ".downsample.1", ds, nullptr, false);
          idt = ds;
        }
        Act o = ebuf(tag + ".out", Ho * Ho, Cout);
        conv2d(pre + ".conv2", t1.p, t1.plane, Ho, Cout, Cout, 3, 1, 1, Ho, false);
        gn(pre + ".bn2", o, &idt, true);
        cur = o;
        Hc = Ho;
        Cc = Cout;
      }
    }
    Act pooled = ebuf("enc.avg", 1, Cc);
    {
      const char* ip = (const char*)cur.p; char* op = (char*)pooled.p;
      const int HW = Hc * Hc, C = Cc;
      const long long ipl = cur.plane, opl = pooled.plane;
      enc_ops.push_back([=, this](int b0, int Bn, int, hipStream_t s) {
        note_other();
        launch_avgpool2d(ip + (size_t)b0 * HW * C * E_, op + (size_t)b0 * C * E_, Bn, HW, C, pr, ipl, opl, s);
      });
    }
    {
      const HostParam& w = P_(R + "fc.weight");
      const float* wd = w.data;
      const int Kf = (int)w.dims[1], Nf = (int)w.dims[0];
      const Packed wp = pack(R + "fc", efmt, Nf, 1, Kf, [=](int n, int, int ci) { return wd[(size_t)n * Kf + ci]; });
      const bool esplit = fmt_split(efmt);
      const int eld = E_ld;
      float* bd;
      if (esplit) {                                        // gemm16 tiles: columns padded to E_ld (zero weights, zero bias)
        std::vector<float> bpad(eld, 0.0f);
        std::memcpy(bpad.data(), P_(R + "fc.bias").data, (size_t)Nf * 4);
        bd = upload_f32(R + "fc.bias#padded", bpad.data(), eld);
      } else {
        bd = vec(R + "fc.bias");
      }
      const char* ip = (const char*)pooled.p;
      float* op = map_emb;
      const long long ppl = pooled.plane;
      enc_ops.push_back([=, this](int b0, int Bn, int, hipStream_t s) {
        ConvGemmParams p{};
        const int Mr = esplit ? (Bn + 255) / 256 * 256 : Bn;      // split: whole tiles (rows beyond Bn are scratch rows)
        p.A = ip + (size_t)b0 * Kf * E_; p.lda = Kf; p.in_Lp = Mr; p.in_stride = 1; p.in_off = 0; p.taps = 1; p.Cin = Kf;
        set_w(p, wp); p.a_plane = ppl;
        p.Out = op + (size_t)b0 * eld; p.ldc = eld; p.out_Lp = Mr; p.out_stride = 1; p.out_off = 0;
        p.L = Mr; p.M = Mr; p.N = esplit ? eld : Nf; p.bias = bd; p.mode = MODE_BIAS; p.out_f32 = 1;
        run_gemm(p, pr, s);
      });
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("sync after build failed");
  // Validation pass: every op of the plan once in DRY-RUN mode (the launchers run their dispatch and shape contracts, nothing is
  // enqueued).  Any layer whose shape fits no kernel (a future kernel_size / n_groups / embedding width) throws HERE, i.e.
  // ditree_denoise_reserve returns an error -- not the first denoise call, and never an abort of the host process.
  {
    float* lm_dev = (float*)dalloc((size_t)bgran * lm * lm * 4);
    lm_ptr = lm_dev;
    denoise_set_dry_run(true);
    try {
      for (auto& op : enc_ops) op(0, bgran, 0, nullptr);
      film_op(bgran, bgran, nullptr);
      for (auto& op : unet_ops) op(bgran, bgran, nullptr);
    } catch (...) {
      denoise_set_dry_run(false);
      lm_ptr = nullptr;
      throw;
    }
    denoise_set_dry_run(false);
    lm_ptr = nullptr;
  }
}

// ------------------------------------------------------------------------------------- glue
static int parse_manifest(DenoiserState* st, const char* manifest, int64_t n_floats) {
  std::istringstream in(manifest);
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    if (line.rfind("#config", 0) == 0) {          // "#config pred_horizon 64 local_map_size 20": what the shapes do not tell
      std::istringstream cs(line.substr(7));
      std::string key;
      int val;
      while (cs >> key >> val) {
        if (key == "pred_horizon") st->P = val;
        else if (key == "local_map_size") st->lm = val;
      }
      continue;
    }
    if (line.rfind("#checksum", 0) == 0) {        // "#checksum <s1 hex> <s2 hex>": Fletcher-style sums over the blob's 32-bit words
      std::istringstream cs(line.substr(9));
      std::string a, b;
      if (!(cs >> a >> b)) return -1;
      const uint32_t* w = (const uint32_t*)st->blob.data();
      uint64_t s1 = 0, s2 = 0;
      for (int64_t i = 0; i < n_floats; ++i) { s1 += w[i]; s2 += s1; }
      if (s1 != std::stoull(a, nullptr, 16) || s2 != std::stoull(b, nullptr, 16)) return -2;
      continue;
    }
    if (line[0] == '#') continue;
    std::istringstream ls(line);
    std::string name;
    int64_t off, n;
    int nd;
    if (!(ls >> name >> off >> n >> nd)) return -1;
    HostParam hp;
    int64_t prod = 1;
    for (int i = 0; i < nd; ++i) {
      int64_t d;
      if (!(ls >> d)) return -1;
      hp.dims.push_back(d);
      prod *= d;
    }
    if (prod != n || off < 0 || off + n > n_floats) return -1;
    hp.data = st->blob.data() + off;
    hp.n = n;
    st->params[name] = hp;
  }
  return 0;
}

static int denoise_core(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx,
                        const float* local_map, const float* cond, int B, int K, const float* t0, const float* dt,
                        const double* act_norm, double* actions, float* x_out, hipStream_t s, float t_scale, int raw,
                        int reuse_encoder);

static int denoise_core_one(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx,
                            const float* local_map, const float* cond, int B, int K, const float* t0, const float* dt,
                            const double* act_norm, double* actions, float* x_out, hipStream_t s, float t_scale, int raw,
                            int reuse_encoder);

// The DDPM branch of the sampler (policies/fm_policy.py:164-182) as K steps inside the library: timesteps[k] is what the
// sinusoidal embedding sees (the scheduler's integer k, NOT scaled by 20), coef (K, 5) the step's sb, sa, c0, c1, sigma,
// z [dev] the standard-normal step noise (row r, step k at r * z_row + k * z_step; rows by noise_idx like the start noise).
int denoise_run_ddpm(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx, const float* local_map,
                     const float* cond, int B, int K, const float* timesteps, const float* coef, const float* z, int64_t z_row,
                     int64_t z_step, const double* act_norm, double* actions, float* x_out, hipStream_t s) {
  DenoiserState* st = ctx->dn;
  if (!st || !st->loaded) return set_err(ctx, DITREE_E_STATE, "denoise: weights not loaded");
  if (!coef || !timesteps || K < 1) return set_err(ctx, DITREE_E_ARG, "denoise_ddpm: schedule missing");
  std::vector<float> ones((size_t)K, 1.0f);
  st->ddpm_coef = coef; st->ddpm_z = z; st->ddpm_z_row = z_row; st->ddpm_z_step = z_step; st->ddpm_row0 = 0;
  const int rc = denoise_core(ctx, noise, noise_stride, noise_idx, local_map, cond, B, K, timesteps, ones.data(), act_norm, actions,
                              x_out, s, 1.0f, 0, 0);
  st->ddpm_coef = nullptr; st->ddpm_z = nullptr;
  return rc;
}

// Candidates per full wave of 256-row x 256-channel tiles over the chip's 256 CUs at the U-Net levels (all levels have the same
// number of tiles: rows halve as channels double): 256 * 65536 / (P * down_dims[0]) -- 512 for the car network (P 64, 512
// channels), 2048 for the ant network (P 16).  What early-exit rounds pack their denoiser calls to.
int denoise_wave_quantum(ditree_ctx* ctx) {
  DenoiserState* st = ctx->dn;
  if (!st || !st->loaded) return 512;
  const long long q = 256ll * 65536ll / ((long long)st->P * st->dims[0]);
  return (int)std::max(64ll, std::min(q, 65536ll));
}

int denoise_run(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx, const float* local_map,
                const float* cond, int B, int K, const float* t0, const float* dt, const double* act_norm, double* actions,
                float* x_out, hipStream_t s) {
  return denoise_core(ctx, noise, noise_stride, noise_idx, local_map, cond, B, K, t0, dt, act_norm, actions, x_out, s,
                      20.0f /* pos_emb_scale, fm_policy.py:187 */, 0, 0);
}

// A call of up to the reserved batch: rows are independent, so it runs as sub-batches of at most the workspace size (every
// pointer advances by the rows done; a compacted round's noise index list advances with them).
static int denoise_core(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx,
                        const float* local_map, const float* cond, int B, int K, const float* t0, const float* dt,
                        const double* act_norm, double* actions, float* x_out, hipStream_t s, float t_scale, int raw,
                        int reuse_encoder) {
  DenoiserState* st = ctx->dn;
  if (!st || !st->loaded) return set_err(ctx, DITREE_E_STATE, "denoise: weights not loaded");
  if (st->prec < 0) return set_err(ctx, DITREE_E_STATE, "denoise: call ditree_denoise_reserve first");
  if (B <= 0 || B > st->Buser) return set_err(ctx, DITREE_E_ARG, "denoise: batch exceeds the reserved workspace");
  if (B <= st->Bmax)
    return denoise_core_one(ctx, noise, noise_stride, noise_idx, local_map, cond, B, K, t0, dt, act_norm, actions, x_out, s,
                            t_scale, raw, reuse_encoder);
  if (!noise || !local_map || !cond) return set_err(ctx, DITREE_E_ARG, "denoise: bad argument");
  const size_t PD = (size_t)st->P * st->D, LM = (size_t)st->lm * st->lm;
  const int row0 = st->ddpm_row0;
  for (int b0 = 0; b0 < B; b0 += st->Bmax) {
    const int bn = std::min(st->Bmax, B - b0);
    st->ddpm_row0 = row0 + (noise_idx ? 0 : b0);
    // (the map embedding of a previous call covers one sub-batch only: larger calls recompute it)
    const int rc = denoise_core_one(ctx, noise_idx ? noise : noise + (size_t)b0 * noise_stride, noise_stride,
                                    noise_idx ? noise_idx + b0 : nullptr, local_map + (size_t)b0 * LM, cond + (size_t)b0 * st->G, bn,
                                    K, t0, dt, act_norm, actions ? actions + (size_t)b0 * PD : nullptr,
                                    x_out ? x_out + (size_t)b0 * PD : nullptr, s, t_scale, raw, 0);
    if (rc) { st->ddpm_row0 = row0; return rc; }
  }
  st->ddpm_row0 = row0;
  return DITREE_OK;
}

// t_scale: factor on t0[k] before the sinusoidal embedding (20 for the flow sampler, 1 for a raw evaluation);
// raw: the last projection writes the network output instead of the Euler update; reuse_encoder: keep the map
// embedding of the previous call (same local maps: the steps of a DDPM loop).
static int denoise_core_one(ditree_ctx* ctx, const float* noise, int64_t noise_stride, const int32_t* noise_idx,
                            const float* local_map, const float* cond, int B, int K, const float* t0, const float* dt,
                            const double* act_norm, double* actions, float* x_out, hipStream_t s, float t_scale, int raw,
                            int reuse_encoder) {
  DenoiserState* st = ctx->dn;
  if (B <= 0 || B > st->Bmax) return set_err(ctx, DITREE_E_ARG, "denoise: batch exceeds the workspace");
  if (!noise || !local_map || !cond || !t0 || !dt || !act_norm || K <= 0 || (!actions && !x_out))
    return set_err(ctx, DITREE_E_ARG, "denoise: bad argument");
  const int Bp = (B + st->bgran - 1) / st->bgran * st->bgran;
  const int uf = st->ufmt;
  {
    const size_t row = (size_t)st->P * st->D * 4;         // one candidate's (P, D) f32 noise
    if (noise_stride < (int64_t)st->P * st->D) return set_err(ctx, DITREE_E_ARG, "denoise: noise stride");
    if (noise_idx)                                       // compacted round: row r takes candidate noise_idx[r]'s noise
      launch_gather_rows_f32(noise, noise_stride, noise_idx, st->x_cur, st->P * st->D, B, s);
    else
      HIP_TRY(ctx, hipMemcpy2DAsync(st->x_cur, row, noise, (size_t)noise_stride * 4, row, (size_t)B,
                                    hipMemcpyDeviceToDevice, s));
  }
  st->lm_ptr = local_map;
  try {
  if (!reuse_encoder) {
    // encoder: up to ENC_SUBS sub-batches on separate streams (fork/join with events on `s`)
    // measured on MI355X (B = 1024): 4 concurrent sub-batches cost more in extra launches than the
    // concurrency returns (34.9 vs 31.6 ms per round); default is one sub-batch, DITREE_ENC_SUBS overrides
    static int subs_env = -1;
    if (subs_env < 0) { const char* e = getenv("DITREE_ENC_SUBS"); subs_env = e ? std::max(1, std::min(atoi(e), (int)DenoiserState::ENC_SUBS)) : 1; }
    const int nsub = (B >= 64 && !fmt_split(st->efmt)) ? subs_env : 1;      // the split encoder's fc writes whole 256-row tiles
    const int per = nsub == 1 ? B : std::min(st->sub_cap, (((B + nsub - 1) / nsub) + 15) / 16 * 16);
    if (nsub > 1 && !st->aux[1]) {
      for (int i = 1; i < DenoiserState::ENC_SUBS; ++i) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&st->aux[i], hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&st->ev_join[i], hipEventDisableTiming));
      }
      HIP_TRY(ctx, hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
    }
    if (nsub > 1) HIP_TRY(ctx, hipEventRecord(st->ev_fork, s));
    for (int i = 0; i < nsub; ++i) {
      const int b0 = i * per, bn = std::min(per, B - b0);
      if (bn <= 0) break;
      hipStream_t si = (i == 0) ? s : st->aux[i];
      if (i > 0) HIP_TRY(ctx, hipStreamWaitEvent(si, st->ev_fork, 0));
      for (auto& op : st->enc_ops) op(b0, bn, i, si);
      if (i > 0) {
        HIP_TRY(ctx, hipEventRecord(st->ev_join[i], si));
        HIP_TRY(ctx, hipStreamWaitEvent(s, st->ev_join[i], 0));
      }
    }
  }
  for (int k = 0; k < K; ++k) {
    const float t = t0[k] * t_scale;
    if (t != st->temb_t) {                                              // batch-invariant: K = 1 computes it once
      launch_time_embed(t, st->dev_f["unet.diffusion_step_encoder.1.weight"], st->dev_f["unet.diffusion_step_encoder.1.bias"],
                        st->dev_f["unet.diffusion_step_encoder.3.weight"], st->dev_f["unet.diffusion_step_encoder.3.bias"],
                        st->temb, s);
      st->temb_t = K == 1 ? t : -1.0f;                                  // several steps share one buffer: recompute
    }
    st->note_other();
    launch_prep_cond(st->temb, st->map_emb, st->E, st->E_ld, cond, st->G, st->condA, B, st->condK, uf, st->cond_plane, s,
                     st->sat_cond);
    st->film_op(B, Bp, s);
    st->note_other();
    launch_prep_sample(st->x_cur, st->named["a0"].p, Bp, st->P, st->D, uf, st->named["a0"].plane, s,
                       st->sat_sample);
    for (auto& op : st->unet_ops) op(B, Bp, s);
    const bool last = (k == K - 1);
    st->note_other();
    FlowStep fs{};
    fs.mode = raw ? 1 : 0;
    fs.dt = dt[k];
    if (st->ddpm_coef) {
      const float* c = st->ddpm_coef + 5 * k;
      fs.mode = 2; fs.sb = c[0]; fs.sa = c[1]; fs.c0 = c[2]; fs.c1 = c[3]; fs.sigma = c[4];
      fs.z = st->ddpm_z ? st->ddpm_z + (long long)k * st->ddpm_z_step : nullptr;
      fs.z_row = st->ddpm_z_row; fs.z_idx = noise_idx; fs.z_row0 = st->ddpm_row0;
      if (fs.sigma != 0.0f && !fs.z) return set_err(ctx, DITREE_E_ARG, "denoise: a DDPM step with sigma > 0 needs step noise");
    }
    launch_final_proj_flow(st->final_h.p, st->final_h.C, st->final_h.Lp(), st->final_h.plane, st->dev_f["unet.final_conv.1.weight"],
                           st->dev_f["unet.final_conv.1.bias"], st->D, st->x_cur, fs, act_norm,
                           (last && actions) ? actions : nullptr, B, st->P, uf, s);
  }
  } catch (const std::exception& e) {
    return set_err(ctx, DITREE_E_ARG, std::string("denoise: ") + e.what());
  }
  if (x_out) HIP_TRY(ctx, hipMemcpyAsync(x_out, st->x_cur, (size_t)B * st->P * st->D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipGetLastError());
  return DITREE_OK;
}

void denoise_destroy(ditree_ctx* ctx) {
  if (ctx->dn) {
    for (auto e : ctx->dn->prof_ev) hipEventDestroy(e);
    ctx->dn->prof_ev.clear();
    for (int i = 1; i < DenoiserState::ENC_SUBS; ++i) {
      if (ctx->dn->aux[i]) hipStreamDestroy(ctx->dn->aux[i]);
      if (ctx->dn->ev_join[i]) hipEventDestroy(ctx->dn->ev_join[i]);
    }
    if (ctx->dn->ev_fork) hipEventDestroy(ctx->dn->ev_fork);
    ctx->dn->free_workspace();
    delete ctx->dn;
    ctx->dn = nullptr;
  }
}

extern "C" {

int32_t ditree_load_weights(ditree_ctx* ctx, const float* blob, int64_t n_floats, const char* manifest, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!blob || !manifest || n_floats <= 0) return set_err(ctx, DITREE_E_ARG, "load_weights: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  denoise_destroy(ctx);
  DenoiserState* st = new (std::nothrow) DenoiserState();
  if (!st) return set_err(ctx, DITREE_E_NOMEM, "load_weights: out of memory");
  st->ctx = ctx;
  try {
    st->blob.assign(blob, blob + n_floats);
    const int prc = parse_manifest(st, manifest, n_floats);
    if (prc == -2) throw std::runtime_error("weight blob does not match the manifest checksum");
    if (prc != 0) throw std::runtime_error("malformed manifest");
    const HostParam& w0 = st->P_("unet.down_modules.0.0.blocks.0.block.0.weight");
    st->D = (int)w0.dims[1];
    for (int i = 0; i < 3; ++i)
      st->dims[i] = (int)st->P_("unet.down_modules." + std::to_string(i) + ".0.blocks.0.block.0.weight").dims[0];
    if (st->has("unet.down_modules.3.0.blocks.0.block.0.weight")) throw std::runtime_error("only 3 U-Net levels supported");
    st->cond_dim = (int)st->P_("unet.down_modules.0.0.cond_encoder.1.weight").dims[1];
    st->E = (int)st->P_("encoder.resnet18.fc.weight").dims[0];
    st->G = st->cond_dim - 256 - st->E;
    if (st->G < 0 || (st->D != 2 && st->D != 8)) throw std::runtime_error("unsupported dimensions (action_dim 2 or 8)");
    if (st->lm != 20 && st->lm != 16) throw std::runtime_error("local_map_size must be 20 (car) or 16 (ant)");
    if (st->P_("unet.diffusion_step_encoder.1.weight").dims[1] != 256) throw std::runtime_error("diffusion_step_embed_dim must be 256");
    for (int i = 0; i < 3; ++i)
      if (st->dims[i] % 64 != 0 || st->dims[i] > 4096) throw std::runtime_error("down_dims must be multiples of 64, <= 4096");
  } catch (const std::exception& e) {
    delete st;
    return set_err(ctx, DITREE_E_ARG, std::string("load_weights: ") + e.what());
  }
  st->loaded = true;
  ctx->dn = st;
  (void)stream;
  return DITREE_OK;
}

int32_t ditree_denoise_reserve(ditree_ctx* ctx, int32_t max_batch, int32_t precision) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!st || !st->loaded) return set_err(ctx, DITREE_E_STATE, "denoise_reserve: weights not loaded");
  if (max_batch <= 0 || precision < DITREE_PREC_BF16 || precision > DITREE_PREC_F16)
    return set_err(ctx, DITREE_E_ARG, "denoise_reserve: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (st->prec == precision && st->Buser >= max_batch) return DITREE_OK;
  try {
    HIP_TRY(ctx, hipDeviceSynchronize());
    st->build(precision, max_batch);
  } catch (const std::exception& e) {
    st->free_workspace();
    return set_err(ctx, DITREE_E_NOMEM, std::string("denoise_reserve: ") + e.what());
  }
  return DITREE_OK;
}

int32_t ditree_denoise(ditree_ctx* ctx, const float* noise, const float* local_map, const float* cond, int32_t B,
                       int32_t K, const float* t0, const float* dt, const double* act_norm, double* actions,
                       float* x_out, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!ctx->dn) return set_err(ctx, DITREE_E_STATE, "denoise: weights not loaded");
  return denoise_run(ctx, noise, (int64_t)ctx->dn->P * ctx->dn->D, nullptr, local_map, cond, B, K, t0, dt, act_norm, actions, x_out,
                     (hipStream_t)stream);
}

int32_t ditree_denoise_ddpm(ditree_ctx* ctx, const float* noise, const float* step_noise, const float* local_map, const float* cond,
                            int32_t B, int32_t K, const float* timesteps, const float* coef, const double* act_norm,
                            double* actions, float* x_out, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!ctx->dn) return set_err(ctx, DITREE_E_STATE, "denoise_ddpm: weights not loaded");
  if (B == 0) return DITREE_OK;
  const int64_t PD = (int64_t)ctx->dn->P * ctx->dn->D;
  return denoise_run_ddpm(ctx, noise, PD, nullptr, local_map, cond, B, K, timesteps, coef, step_noise, (int64_t)K * PD, PD, act_norm,
                          actions, x_out, (hipStream_t)stream);
}

int32_t ditree_denoise_eval(ditree_ctx* ctx, const float* sample, const float* local_map, const float* cond, int32_t B,
                            float timestep, int32_t reuse_encoder, float* out, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  if (!ctx->dn) return set_err(ctx, DITREE_E_STATE, "denoise_eval: weights not loaded");
  if (!out) return set_err(ctx, DITREE_E_ARG, "denoise_eval: bad argument");
  const float one = 1.0f;
  double unit[16];
  for (int d = 0; d < 8; ++d) { unit[d] = 0.0; unit[8 + d] = 1.0; }
  for (int d = 0; d < ctx->dn->D && d < 8; ++d) { unit[d] = 0.0; unit[ctx->dn->D + d] = 1.0; }
  return denoise_core(ctx, sample, (int64_t)ctx->dn->P * ctx->dn->D, nullptr, local_map, cond, B, 1, &timestep, &one, unit,
                      nullptr, out, (hipStream_t)stream, 1.0f, 1, reuse_encoder);
}

int32_t ditree_denoise_dims(ditree_ctx* ctx, int32_t* dims5) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!st || !st->loaded || !dims5) return set_err(ctx, DITREE_E_STATE, "denoise_dims: weights not loaded");
  dims5[0] = st->P; dims5[1] = st->D; dims5[2] = st->lm; dims5[3] = st->G; dims5[4] = st->E;
  return DITREE_OK;
}

int32_t ditree_denoise_status(ditree_ctx* ctx, int32_t* n_saturated, char* names, int64_t names_cap, int32_t clear,
                              void* stream) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!n_saturated || (names_cap > 0 && !names)) return set_err(ctx, DITREE_E_ARG, "denoise_status: bad argument");
  *n_saturated = 0;
  if (names_cap > 0) names[0] = '\0';
  if (!st || st->prec < 0 || st->sat_flags == nullptr || st->sat_names.empty()) return DITREE_OK;   // nothing can saturate
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> host(st->sat_names.size());
  HIP_TRY(ctx, hipMemcpyAsync(host.data(), st->sat_flags, host.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  std::string text;
  for (size_t i = 0; i < host.size(); ++i)
    if (host[i] != 0) {
      ++*n_saturated;
      if (!text.empty()) text += "\n";
      text += st->sat_names[i];
    }
  if (names_cap > 0) {
    const size_t n = std::min((size_t)names_cap - 1, text.size());
    std::memcpy(names, text.data(), n);
    names[n] = '\0';
  }
  if (clear && *n_saturated) HIP_TRY(ctx, hipMemsetAsync(st->sat_flags, 0, host.size() * sizeof(int), s));
  return DITREE_OK;
}

int32_t ditree_profile(ditree_ctx* ctx, int32_t enable) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!st) return set_err(ctx, DITREE_E_STATE, "profile: weights not loaded");
  st->prof_collect();
  st->prof_on = enable != 0;
  st->prof_mode = enable;
  if (enable) for (int k = 0; k < 3; ++k) { st->prof_ms_done[k] = 0.0; st->prof_launches_done[k] = 0; st->prof_flops_done[k] = 0.0; }
  return DITREE_OK;
}

int32_t ditree_profile_read(ditree_ctx* ctx, double* ms3, int64_t* launches3, double* flops3) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!st || !ms3 || !launches3 || !flops3) return set_err(ctx, DITREE_E_STATE, "profile_read: bad state");
  st->prof_collect();
  for (int k = 0; k < 3; ++k) {
    ms3[k] = st->prof_ms_done[k];
    launches3[k] = st->prof_launches_done[k];
    flops3[k] = st->prof_flops_done[k];
  }
  return DITREE_OK;
}

int32_t ditree_denoise_debug_read(ditree_ctx* ctx, const char* name, int32_t B, float* out, int64_t capacity,
                                  int32_t* dims3, void* stream) {
  if (!ctx) return DITREE_E_ARG;
  DenoiserState* st = ctx->dn;
  if (!st || st->prec < 0 || !name || !out || !dims3) return set_err(ctx, DITREE_E_STATE, "debug_read: no workspace");
  hipStream_t s = (hipStream_t)stream;
  std::string nm(name);
  if (nm == "film" || nm == "map_emb") {
    const int cols = nm == "film" ? st->film_cols : st->E;
    const int ld = nm == "film" ? st->film_cols : st->E_ld;
    const float* src = nm == "film" ? st->film : st->map_emb;
    if ((int64_t)B * cols > capacity) return set_err(ctx, DITREE_E_ARG, "debug_read: capacity");
    HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)cols * 4, src, (size_t)ld * 4, (size_t)cols * 4, (size_t)B, hipMemcpyDeviceToDevice, s));
    dims3[0] = B; dims3[1] = 1; dims3[2] = cols;
    return DITREE_OK;
  }
  auto it = st->named.find(nm);
  if (it == st->named.end()) return set_err(ctx, DITREE_E_ARG, "debug_read: unknown buffer " + nm);
  const Act& a = it->second;
  if ((int64_t)B * a.L * a.C > capacity) return set_err(ctx, DITREE_E_ARG, "debug_read: capacity");
  launch_unpack_act(a.p, a.ld, a.coff, a.Lp(), a.padded ? 1 : 0, out, B, a.L, a.C, a.fmt, a.plane, s);
  dims3[0] = B; dims3[1] = a.L; dims3[2] = a.C;
  HIP_TRY(ctx, hipGetLastError());
  return DITREE_OK;
}

}  // extern "C"
