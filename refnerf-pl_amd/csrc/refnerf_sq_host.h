/*
 * refnerf_sq_host.h -- host-side seam between the two translation units of librefnerf_hip.so:
 * refnerf_hip.hip (C ABI, every kernel of rounds 1-4) and refnerf_sq_train.hip (the round-5 training kernels of the
 * parity-grade 16-bit mode: refnerf_sq_layout.h).  Internal: nothing here is part of include/refnerf_hip.h.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "refnerf_hip.h"

namespace rnh {   /* defined in refnerf_hip.hip */
int fail(int code, const char *fmt, const char *detail = "");
int timer_begin(hipStream_t st, long *slot, int fam);
int timer_end(hipStream_t st, long slot);
bool prof_on();
int prof_buffer(long long **out);
int lds_pad();
}  // namespace rnh

namespace rnsq {  /* defined in refnerf_sq_train.hip */
size_t image_bytes();
int pack(const float *d_params, void *d_packed, hipStream_t st);
/* the training forward of one level (REFNERF_PREC_F16X2, built-in basis); d_act: REFNERF_ACT_SQ */
int forward(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays, int R, const float *d_sdist_in,
            const float *d_weights_in, const refnerf_level_out *out, float *d_act, hipStream_t st);
/* the per-sample backward (the seeds are already in `d_seeds`): writes DELTA + the factor units */
int backward_chain(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays, int R, const float *d_sdist,
                   const refnerf_level_grads *grads, const float *d_act, float *d_delta, const float *d_seeds, long long pitch,
                   hipStream_t st);
/* dW partials of the level: PART[*slices_used][NUM_PARAMS], *slices_used <= slices (the caller reduces them) */
int wgrad(const float *d_act, const float *d_delta, long long S, long long pitch, int k_per_slice, int slices, float *d_part,
          float *d_kmin, int act11, int *slices_used, hipStream_t st);
}  // namespace rnsq
