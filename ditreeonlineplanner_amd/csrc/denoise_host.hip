// placeholder: replaced by the MFMA denoiser host code
#include "ditree_internal.h"
int denoise_run(ditree_ctx* ctx, const float*, const float*, const float*, int, int, const float*, const float*,
                const double*, double*, float*, hipStream_t) {
  return set_err(ctx, DITREE_E_STATE, "denoiser weights not loaded");
}
void denoise_destroy(ditree_ctx*) {}
extern "C" {
int32_t ditree_load_weights(ditree_ctx* ctx, const float*, int64_t, const char*, void*) {
  return set_err(ctx, DITREE_E_STATE, "not built");
}
int32_t ditree_denoise_reserve(ditree_ctx* ctx, int32_t, int32_t) { return set_err(ctx, DITREE_E_STATE, "not built"); }
int32_t ditree_denoise(ditree_ctx* ctx, const float*, const float*, const float*, int32_t, int32_t, const float*,
                       const float*, const double*, double*, float*, void*) {
  return set_err(ctx, DITREE_E_STATE, "not built");
}
}
