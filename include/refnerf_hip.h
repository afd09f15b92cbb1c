/*
 * refnerf_hip.h -- C ABI of the MI355X (gfx950) Ref-NeRF rendering inner loop.
 *
 * Shared library: refnerf-pl_amd/csrc/librefnerf_hip.so.  Plain pointers and
 * sizes only; every `d_*` pointer is DEVICE memory owned by the caller; every
 * call is asynchronous on `stream` (a hipStream_t passed as void*, NULL = the
 * default stream) and returns 0 or a negative REFNERF_E* code (message via
 * refnerf_last_error()).  Level / stage calls keep no state between calls and
 * may be issued concurrently from several host threads (each on its own
 * stream); the error string is per thread.  The only process-wide state is the
 * opt-in kernel timer of refnerf_set_timing / refnerf_get_timing (mutex-guarded)
 * and two debug knobs read once from the environment (REFNERF_PROF,
 * REFNERF_LDS_PAD).
 *
 * The upstream reference (minfenli/refnerf-pl) has no FFI: its boundary is the
 * Python call surface of internal/models.py.  Each entry point below names the
 * reference function (file:line, relative to the upstream repo root) whose
 * ATen op sequence it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 */
#ifndef REFNERF_HIP_H
#define REFNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REFNERF_ABI_VERSION 11  /* v11: refnerf_level_image (which weight image a level configuration expects as d_packed) + the library remembers the kind of every image it packs and refuses a mismatching d_packed; v10: REFNERF_IMAGE_F16X2_TRAIN + REFNERF_ACT_SQ (training levels of REFNERF_PREC_F16X2 on the eval kernel's skeleton: d_packed of refnerf_level_forward_train / refnerf_level_backward is the train image then); v9: REFNERF_ACT_F16X2 (split-f16 ACT / DELTA formats of the split-f16 training chains), refnerf_activations_format; v8: cfg.ipe_groups, refnerf_pack_weights_basis (general IPE bases: icosahedron); v7: REFNERF_PREC_F16X2 (split-operand f16: the parity-grade 16-bit inference mode); v6: cfg.dir_enc (REFNERF_DIRENC_*), cfg.raydist (REFNERF_RAYDIST_*), cfg.disable_integration; v5: refnerf_render_rays, REFNERF_PREC_F16, refnerf_get_timing_family, refnerf_losses_forward / _backward; v4: cfg.wgrad_mode, refnerf_level_saved.activations_format, bf16-chain training modes */
#define REFNERF_NUM_PARAMS 1110158 /* canonical fp32 blob, nerf_mlp.* state_dict order */

enum {
  REFNERF_OK = 0,
  REFNERF_EINVAL = -1,      /* bad argument / unsupported configuration     */
  REFNERF_ENOGPU = -2,      /* no gfx950 device                             */
  REFNERF_EHIP = -3,        /* HIP runtime error                            */
  REFNERF_EUNSUPPORTED = -4 /* valid in the reference, not built here yet   */
};

/* arithmetic of the MLP contractions */
enum {
  REFNERF_PREC_F32 = 0,  /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains (parity mode) */
  REFNERF_PREC_BF16 = 1, /* v_mfma_f32_32x32x16_bf16, fp32 accumulate (training forward / backward in
                            this mode: n_samples <= 294, REFNERF_EINVAL beyond -- LDS budget)  */
  REFNERF_PREC_F16 = 2,  /* v_mfma_f32_32x32x16_f16 (IEEE half operands, fp32 accumulate): the bf16 inference kernel
                            with 11 instead of 8 significand bits -- 8-10x closer to REFNERF_PREC_F32 at ~3 % lower
                            throughput; hidden activations must stay below 65504.  refnerf_level_forward only. */
  REFNERF_PREC_F16X2 = 3 /* split-operand f16: in the spatial trunk and the density / scalar head block BOTH operands are
                            hi + lo pairs of IEEE halves (22 significand bits; the partial products hi*hi + lo*hi + hi*lo
                            on v_mfma_f32_16x16x32_f16, fp32 accumulate; lo*lo = 2^-22 of a product is dropped), the
                            directional trunk is plain f16; resampler,
                            encodings, activations and compositing are the fp32 parity code.  The 16-bit mode that holds
                            the reference's fp32 nn.Linear arithmetic (internal/models.py:576-580, 686-700) on trained
                            weights: every ray within max(1e-4, 4 x the reference's own fp32 rounding error) of the float64
                            value of the same function (131 k rays; worst ray 1.8e-4 from the fp32 reference, 1.6e-4 from
                            float64 where the reference itself is 1.6e-4 off; within 1e-4 of the fp32 reference on all but <= 2
                            rays of a batch of 8192: tests/test_hip_f16x2.py);
                            ~1.5x the matrix cycles of REFNERF_PREC_F16.  Range: the hi halves are IEEE
                            halves, so weights and hidden activations must stay below 65504 in magnitude (a trained
                            Ref-NeRF's reach ~1e2; checked up to 8e3: 1e-6).  Beyond it a unit becomes hi = inf, lo = -inf, the
                            next layer's accumulators NaN; the ReLUs of the split kernels (inference, training chains, general
                            basis) let a NaN through, so the ray's outputs are NaN: loud, never a finite wrong colour.
                            REFNERF_PREC_F32 has no such limit.
                            refnerf_level_forward_train / refnerf_level_backward in this mode, built-in IPE basis (v10):
                            the eval kernel's skeleton on the REFNERF_IMAGE_F16X2_TRAIN image -- spatial trunk and
                            density-normal VJP W_hi x_hi + W_lo x_hi + W_hi x_lo, directional trunk [W_hi | W_lo] x on one
                            half of x, backward W^T (22 bits) x delta (11 bits); saved activations REFNERF_ACT_SQ (then the
                            backward must run in this mode too).  General basis: the f32 skeleton with three-product
                            contractions, REFNERF_ACT_F32.  MEASURED gradient rel-L2 from the reference's autograd (GPU
                            fixtures, tests/test_hip_f16x2.py): 3.5e-5 (shiny) .. 8.1e-5 (llff) .. 1.1e-4 (trained-like),
                            1.6e-4 (icosahedron basis); trained_long 1.5e-3 in EVERY chain mode incl. f32 (level-1 sample
                            positions on a sharp surface, not arithmetic); between the chain modes from identical step
                            functions 7e-5 .. 2.7e-4.  Strict gradient parity = REFNERF_PREC_F32 chains; 22-bit deltas on
                            fp32 rows = f32 forward + this mode's backward (level_bwd_f16x2c_r32: 2e-6 between the modes). */
};

/* arithmetic of the weight-gradient GEMM of refnerf_level_backward (dW = DELTA x ACT^T over the samples) */
enum {
  REFNERF_WGRAD_F32 = 0,    /* v_mfma_f32_32x32x2_f32: fp32 products                                     */
  REFNERF_WGRAD_BF16X3 = 1, /* the 16-bit-MFMA GEMM that goes with the chains (default).  After f32 chains: operands split hi + lo into
                               bf16 pairs, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate: 2^-16 per product,
                               HBM-bound.  After REFNERF_PREC_F16X2 chains (REFNERF_ACT_SQ): the f16 GEMM on the saved halves --
                               spatial layer inputs hi + lo (22 bits), directional layer inputs and every delta ONE half.
                               After bf16 chains: bf16 rows. */
  REFNERF_WGRAD_F16 = 2     /* v10, REFNERF_PREC_F16X2 training levels on the built-in basis only (set it in the cfg of BOTH
                               refnerf_level_forward_train and refnerf_level_backward): the spatial layer inputs at ONE half as
                               well -- the forward does not write their lo halves, the GEMM does not read them: 17.4 instead of
                               21.7 KB of weight-gradient operands per ray-sample, one MFMA product per tile.  A layer input's
                               rounding is independent per sample, so it averages over the batch where a weight's does not:
                               scripts/exp_train_sq_precision.py measures +1 .. 5 % on the gradient's distance from the
                               reference's autograd (three trained weight sets). */
};

/* element type of the saved layer inputs: fp32 rows (f32 training forward; any forward with a general IPE basis), bf16 rows
 * (bf16-chain training forward, whose activations are bf16-exact: half the stream; rows 2j and 2j+1 share the dwords of
 * pair-row j, low / high half) or split-f16 pair units (REFNERF_ACT_F16X2, the split-f16 training forward on the built-in
 * basis: per pair of rows one dword of packed hi halves and one of packed lo halves, x = hi + lo -- the chain kernels' own
 * B fragments, stored without arithmetic; the split-f16 backward then writes its layer deltas as ONE half per element plus a
 * power-of-two factor per (layer, sample), and the weight-gradient GEMM runs on v_mfma_f32_32x32x16_f16: 26 instead of
 * 34.5 KB of operands per ray-sample).  The buffer is opaque to the caller, only its size is part of the ABI; the 128 rows of
 * ReLU mask words are 32-bit in all three.  refnerf_activations_format(cfg) tells which one refnerf_level_forward_train
 * writes for a configuration: pass it on in refnerf_level_saved.activations_format. */
enum { REFNERF_ACT_F32 = 0, REFNERF_ACT_BF16 = 1, REFNERF_ACT_F16X2 = 2,
       REFNERF_ACT_SQ = 3 /* v10: what the REFNERF_PREC_F16X2 training forward writes on the built-in basis: spatial layer inputs as
                             hi / lo pair units, directional layer inputs as the ONE half their trunk multiplies, lane-local ReLU
                             sign words, the raw scalar head rows and raw rgb (13.7 KB per ray-sample; the buffer keeps the size
                             refnerf_activation_workspace_bytes reports); the backward writes its deltas as one half per element
                             + two power-of-two factors per (layer, sample): 22 KB of weight-gradient operands per ray-sample */ };

/* a weight image that is not an arithmetic mode of refnerf_level_forward: refnerf_pack_weights / refnerf_packed_weights_bytes
 * take it in place of a REFNERF_PREC_* code */
enum {
  REFNERF_IMAGE_F16X2_TRAIN = 4 /* v10: forward + transposed operands of the REFNERF_PREC_F16X2 TRAINING kernels (built-in IPE
                                   basis) as one stream of 17 KB chunks (11.5 MB): d_packed of refnerf_level_forward_train and
                                   refnerf_level_backward when cfg.precision = REFNERF_PREC_F16X2 and cfg.ipe_groups <= 1
                                   (a general basis and the other precisions keep the REFNERF_PREC_F32 image) */
};

/* encoding of the (reflected) direction fed to the directional MLP (internal/models.py:484-492) */
enum {
  REFNERF_DIRENC_IDE = 0,    /* MLP.use_directional_enc = True: ref_utils.generate_ide_fn(5), 72 features                 */
  REFNERF_DIRENC_POSENC = 1  /* False: coord.pos_enc(d, 0, 5, append_identity) (coord.py:136-147), its 33 features in the
                                slots [x y z | sin(2^j d_i) j-major | 0 x18 || sin(2^j d_i + pi/2) | 0 x21] of the same 72-wide
                                block (the caller embeds the [.., 33] weight columns there; the roughness is not used)   */
};

/* Model.raydist_fn: the bijection between normalised and metric ray distance (coord.construct_ray_warps, coord.py:63-99):
 * t = fn_inv(s * fn(far) + (1 - s) * fn(near)) */
enum {
  REFNERF_RAYDIST_NONE = 0,        /* fn = None: linear in t                                   */
  REFNERF_RAYDIST_PIECEWISE = 1,   /* 'piecewise': x < 1 ? x / 2 : 1 - 1 / (2 x) (allows near = 0) */
  REFNERF_RAYDIST_RECIPROCAL = 2,  /* torch.reciprocal (linear in disparity)                   */
  REFNERF_RAYDIST_LOG = 3, REFNERF_RAYDIST_EXP = 4, REFNERF_RAYDIST_SQRT = 5, REFNERF_RAYDIST_SQUARE = 6
};

enum { REFNERF_SRGB_NONE = 0, REFNERF_SRGB_LINEAR = 1, REFNERF_SRGB_NORM_LINEAR = 2,
       REFNERF_SRGB_SRGB = 3, REFNERF_SRGB_NORM_SRGB = 4 };

/* One sampling level of Model.__call__ (internal/models.py:162-306).
 * Field-for-field the knobs the reference reads on this path. */
typedef struct refnerf_level_cfg {
  int32_t n_samples;          /* Model.num_prop_samples / num_nerf_samples (models.py:164) */
  int32_t n_in;               /* intervals of the incoming step function (1 at level 0)    */
  int32_t training;           /* MLP.training: also emit density-gradient normals (:603)   */
  int32_t compute_extras;     /* models.py:133                                             */
  int32_t srgb_mapping;       /* MLP.srgb_mapping (:712)                                   */
  int32_t srgb_mapping_normalization; /* (:718)                                           */
  int32_t render_srgb_mode;   /* Config.srgb_mapping_type if srgb_mapping_when_rendering (:285-287) */
  int32_t opaque_background;  /* Model.opaque_background (render.py:139-143)               */
  int32_t ray_shape;          /* 0 'cone', 1 'cylinder' (render.py:121-126)                */
  int32_t precision;          /* REFNERF_PREC_*                                            */
  int32_t wgrad_mode;         /* REFNERF_WGRAD_*: arithmetic of the weight-gradient GEMM (backward only) */
  int32_t dir_enc;            /* REFNERF_DIRENC_*: MLP.use_directional_enc (:484-492)      */
  int32_t raydist;            /* REFNERF_RAYDIST_*: Model.raydist_fn (:147, coord.py:63-99) */
  int32_t disable_integration;/* Model.disable_integration (:228-231): zero covariances -> plain positional encoding */
  int32_t ipe_groups;         /* MLP.basis_shape / basis_subdivisions (:384-385, 482-484): 0 or 1 = the octahedron/1 basis the kernels are
                                 built around; G = 2..7: 3 G basis directions, d_packed from refnerf_pack_weights_basis */
  float anneal;               /* models.py:190-195                                         */
  float resample_padding;     /* Model.resample_padding (:202)                             */
  float s_near, s_far;        /* Model.init_s_near / init_s_far (:213)                     */
  float density_bias;         /* MLP.density_bias (:623)                                   */
  float roughness_bias;       /* MLP.roughness_bias (:641)                                 */
  float rgb_premultiplier, rgb_bias, rgb_padding; /* (:700, :729)                          */
  float bg_rgb;               /* Model.bg_intensity_range midpoint (:261-267)              */
} refnerf_level_cfg;

void refnerf_level_cfg_default(refnerf_level_cfg *cfg);

/* utils.Rays (internal/utils.py:51-93): the six fields the path reads. */
typedef struct refnerf_rays {
  const float *d_origins;    /* [R,3] */
  const float *d_directions; /* [R,3] */
  const float *d_viewdirs;   /* [R,3] */
  const float *d_radii;      /* [R]   */
  const float *d_near;       /* [R]   */
  const float *d_far;        /* [R]   */
} refnerf_rays;

/* Outputs of one level; any pointer may be NULL (not stored).
 * Per-sample = ray_history[l] (models.py:731-750, 304-305),
 * per-ray = renderings[l] (render.py:152-254). */
typedef struct refnerf_level_out {
  float *d_sdist;        /* [R,N+1] */
  int32_t *d_bin_idx;    /* [R,N]   CDF bin of each sample centre (math.py:93) */
  float *d_density;      /* [R,N]   */
  float *d_rgb;          /* [R,N,3] */
  float *d_normals;      /* [R,N,3] training only */
  float *d_normals_pred; /* [R,N,3] */
  float *d_grad_pred;    /* [R,N,3] */
  float *d_tint;         /* [R,N,3] */
  float *d_diffuse;      /* [R,N,3] */
  float *d_specular;     /* [R,N,3] */
  float *d_roughness;    /* [R,N]   */
  float *d_weights;      /* [R,N]   */
  float *d_r_rgb, *d_r_diffuse, *d_r_specular; /* [R,3] */
  float *d_r_distance;   /* [R] */
  float *d_r_acc;        /* [R] */
  float *d_r_normals, *d_r_normals_pred, *d_r_tint; /* [R,3] extras */
  float *d_r_roughness;  /* [R] */
  float *d_r_distance_mean; /* [R] */
  double *d_r_percentiles;  /* [R,3] float64: 5 / 50 / 95 (math.py:133-135) */
} refnerf_level_out;

/* Library / device probe.  refnerf_device_ok() returns REFNERF_OK only when
 * the current HIP device is gfx950. */
int refnerf_abi_version(void);
int refnerf_device_ok(void);
const char *refnerf_last_error(void);

/* Bytes of the kernel-side weight image for a precision mode. */
size_t refnerf_packed_weights_bytes(int precision);

/* Re-layout the 46 nn.Parameters of NerfMLP (canonical blob, device) into the
 * MFMA operand image the level kernel streams.  Replaces nothing in the
 * reference (its weights are consumed in place by nn.Linear, models.py:576-
 * 700); must be re-run after every optimiser step. */
int refnerf_pack_weights(const float *d_params, void *d_packed, int precision, void *stream);

/* The same for an MLP whose integrated positional encoding projects onto a general basis (geopoly.generate_basis,
 * internal/geopoly.py:78-123; coord.lift_and_diagonalize, internal/coord.py:129-133; NerfMLP's constructor default is
 * the 21 directions of 'icosahedron' / 2): d_basis [3 * ipe_groups][3] = the basis rows in the reference's order and
 * component order, ipe_groups <= 7.  d_params is then REFNERF_NUM_PARAMS_EXT floats: the canonical blob, whose IPE
 * columns of spatial_net.0 / .5 belong to directions 0..2, followed by W_ext[layer: 0, 5][256 rows][576]: column
 * 96 (g - 1) + 48 c + 3 j + b = (direction group g = 1..6, sin | cos block c, degree j, direction 3 g + b); unused
 * groups zero.  The image is built for REFNERF_PREC_F32 only (levels then run with cfg.precision REFNERF_PREC_F32 or
 * REFNERF_PREC_F16X2 on it: it carries the split-f16 copies); it is
 * refnerf_packed_weights_bytes_basis(precision, ipe_groups) bytes and levels run with cfg.ipe_groups = ipe_groups. */
#define REFNERF_NUM_PARAMS_EXT (REFNERF_NUM_PARAMS + 2 * 6 * 256 * 96)
size_t refnerf_packed_weights_bytes_basis(int precision, int ipe_groups);
int refnerf_pack_weights_basis(const float *d_params, const float *d_basis, int ipe_groups, void *d_packed, int precision, void *stream);

/* One fused launch = one iteration of the level loop of Model.__call__
 * (models.py:162-306): resample (stepfun.py:209-258) -> s_to_t (coord.py:96-98)
 * -> cast_rays (render.py:105-129) -> MLP.__call__ (models.py:533-750) ->
 * compute_alpha_weights (render.py:132-149) -> volumetric_rendering
 * (render.py:152-254).  d_sdist_in [R,n_in+1], d_weights_in [R,n_in]. */
int refnerf_level_forward(const void *d_packed, const refnerf_level_cfg *cfg,
                          const refnerf_rays *rays, int32_t R,
                          const float *d_sdist_in, const float *d_weights_in,
                          const refnerf_level_out *out, void *stream);

/* ---- training (the autograd graph of the same level) ----
 * With cfg->training = 1 (f32 precision mode) refnerf_level_forward also emits
 * the density-gradient normals (models.py:603-609).  refnerf_level_backward is
 * what `loss.backward()` runs for one level in the reference
 * (internal/nerf_system.py training_step -> autograd through models.py:162-306):
 * given the level's saved forward outputs and dL/d(outputs) for the outputs the
 * reference's losses read (train_utils.py:33-204: rendering rgb, history
 * weights, history normals_pred; plus rendering acc / distance, which are
 * linear in the weights; the density-gradient normals, sdist and the
 * resampling inputs are detached there as well), it ACCUMULATES dL/d(params)
 * into d_param_grads (canonical blob, REFNERF_NUM_PARAMS floats).
 * d_workspace holds the per-sample output gradients of every layer, which the
 * weight-gradient GEMM contracts with the saved layer inputs, the split-K
 * partial sums and the per-sample seeds of the per-ray pre-pass. */
typedef struct refnerf_level_saved {
  const float *d_sdist;     /* [R,N+1] refnerf_level_out.d_sdist   */
  const float *d_density;   /* [R,N]                               */
  const float *d_rgb;       /* [R,N,3]                             */
  const float *d_weights;   /* [R,N]                               */
  const void *d_activations; /* the buffer refnerf_level_forward_train filled */
  int32_t activations_format; /* REFNERF_ACT_*: how that forward wrote it (its cfg->precision)              */
} refnerf_level_saved;

typedef struct refnerf_level_grads {
  const float *d_g_r_rgb;         /* [R,3]   dL/d renderings['rgb']                  */
  const float *d_g_weights;       /* [R,N]   dL/d ray_history['weights'], or NULL    */
  const float *d_g_normals_pred;  /* [R,N,3] dL/d ray_history['normals_pred'], or NULL */
  const float *d_g_r_acc;         /* [R]     dL/d renderings['acc'], or NULL             */
  const float *d_g_r_distance;    /* [R]     dL/d renderings['distance'], or NULL        */
  /* ABI v3: optional per-sample seeds on the remaining differentiable entries of ray_history
   * (internal/models.py:731-750), consumed by the regularisers of internal/train_utils.py:207-325.
   * Per-RAY seeds on renderings['diffuse' / 'specular' / 'normals' / 'normals_pred' / 'tint' /
   * 'roughness'] are linear in (weights, history) -- render.py:161-165,227-231 -- and are folded into
   * d_g_weights and these per-sample seeds by the host (refnerf-pl_amd/models.py::_fold_ray_seeds). */
  const float *d_g_density;       /* [R,N]   dL/d ray_history['density'], or NULL           */
  const float *d_g_rgb;           /* [R,N,3] dL/d ray_history['rgb'], or NULL               */
  const float *d_g_diffuse;       /* [R,N,3] dL/d ray_history['diffuse'], or NULL           */
  const float *d_g_specular;      /* [R,N,3] dL/d ray_history['specular'], or NULL          */
  const float *d_g_tint;          /* [R,N,3] dL/d ray_history['tint'], or NULL              */
  const float *d_g_roughness;     /* [R,N]   dL/d ray_history['roughness'], or NULL         */
} refnerf_level_grads;

/* Training forward that also keeps every linear layer's input for the backward
 * (what autograd saves for nn.Linear in the reference): d_activations is a
 * caller-owned buffer of refnerf_activation_workspace_bytes(R, N) bytes
 * (18.1 KB per ray-sample) that must stay untouched until the level's
 * refnerf_level_backward has run.  cfg->training must be 1.  d_packed is the REFNERF_PREC_F32 image;
 * cfg->precision = REFNERF_PREC_BF16 runs the MLP chains (and the density-normal VJP) on bf16 MFMA with the
 * activations rounded to bf16 once per layer (RGB within 1e-4 of the f32 mode); REFNERF_PREC_F32 is the parity mode. */
size_t refnerf_activation_workspace_bytes(int32_t R, int32_t n_samples);
int refnerf_activations_format(const refnerf_level_cfg *cfg);   /* REFNERF_ACT_* of refnerf_level_forward_train(cfg), -1 for NULL */
/* v11: the weight image refnerf_level_forward / _forward_train / _backward expect as `d_packed` for this configuration -- the
 * `precision` argument to pass to refnerf_pack_weights (REFNERF_PREC_* or REFNERF_IMAGE_F16X2_TRAIN; a general IPE basis:
 * REFNERF_PREC_F32 through refnerf_pack_weights_basis), -1 for NULL.  One rule, inside the library (it depends on
 * cfg->training, cfg->precision, cfg->ipe_groups and on the REFNERF_LEGACY_F16X2_TRAIN switch the library itself reads).
 * The library also remembers, per device pointer, the kind of every image refnerf_pack_weights* wrote: a level entry handed a
 * pointer it packed as ANOTHER kind returns REFNERF_EINVAL instead of streaming garbage (pointers it never packed -- a
 * caller's own copy of an image -- are taken on trust). */
int refnerf_level_image(const refnerf_level_cfg *cfg);
int refnerf_level_forward_train(const void *d_packed, const refnerf_level_cfg *cfg,
                                const refnerf_rays *rays, int32_t R,
                                const float *d_sdist_in, const float *d_weights_in,
                                const refnerf_level_out *out, void *d_activations,
                                size_t activations_bytes, void *stream);

/* Scratch of refnerf_level_backward (per-layer output gradients + split-K partials; 17 KB per ray-sample).
 * refnerf_level_backward: d_packed is the REFNERF_PREC_F32 image of refnerf_pack_weights; cfg->precision selects the
 * arithmetic of the transposed GEMM chains (REFNERF_PREC_F32: exact fp32 MFMA, the parity mode; REFNERF_PREC_BF16:
 * bf16 MFMA with deltas rounded to bf16 once per layer, gradients within ~1e-3 relative L2 of the f32 mode),
 * cfg->wgrad_mode that of the weight-gradient GEMM. */
size_t refnerf_backward_workspace_bytes(int32_t R, int32_t n_samples);
/* ... and of both buffers for a level with a general IPE basis (cfg.ipe_groups > 1: the IPE features of the extra direction
 * groups behind the activations, the tail's partial sums behind the workspace; d_param_grads is then
 * REFNERF_NUM_PARAMS_EXT floats, the gradient of the tail columns of groups the basis does not have -- group index >=
 * ipe_groups -- is exactly 0; f32 / split-f16 chains and the bf16x3 weight-gradient GEMM) */
size_t refnerf_backward_workspace_bytes_basis(int32_t R, int32_t n_samples, int32_t ipe_groups);
size_t refnerf_activation_workspace_bytes_basis(int32_t R, int32_t n_samples, int32_t ipe_groups);

int refnerf_level_backward(const void *d_packed, const refnerf_level_cfg *cfg,
                           const refnerf_rays *rays, int32_t R,
                           const refnerf_level_saved *saved, const refnerf_level_grads *grads,
                           float *d_param_grads, void *d_workspace, size_t workspace_bytes,
                           void *stream);

/* MLP.__call__ (internal/models.py:533-750) on caller-supplied Gaussians -- the
 * reference's per-sample entry point: d_means [R,N,3], d_covs [R,N,3,3]
 * (cov_is_full) or [R,N,3] (diagonal), d_viewdirs [R,3] -> the `ray_results`
 * dict as the per-sample fields of the refnerf_level_out struct: density, rgb, normals
 * when cfg->training, normals_pred, grad_pred, tint, diffuse, specular,
 * roughness; the other pointers are ignored.  Same device code as the fused
 * level kernel without the resampling and compositing phases; f32 mode. */
int refnerf_mlp_forward(const void *d_packed, const refnerf_level_cfg *cfg,
                        const float *d_means, const float *d_covs, int32_t cov_is_full,
                        const float *d_viewdirs, int32_t R, int32_t N,
                        const refnerf_level_out *out, void *stream);

/* The step in front of the path, on the device: camera_utils.pixels_to_rays
 * (internal/camera_utils.py:502-614) for perspective cameras without lens
 * distortion, with the optional NDC conversion (:31-97, near = 1).
 * d_pix_x / d_pix_y [n] int32; d_pixtocams [3,3] (or [n,3,3] when
 * pixtocam_per_ray); d_camtoworlds [3,4] (or [n,3,4]); d_pixtocam_ndc [3,3] or
 * NULL.  Outputs: origins / directions / viewdirs [n,3], radii [n],
 * imageplane [n,2] (optional).  Saves the 64 B/ray host-to-device copy and the
 * numpy ray casting of whole-image rendering. */
int refnerf_pixels_to_rays(const int32_t *d_pix_x, const int32_t *d_pix_y,
                           const float *d_pixtocams, int32_t pixtocam_per_ray,
                           const float *d_camtoworlds, int32_t camtoworld_per_ray,
                           const float *d_pixtocam_ndc, int32_t n,
                           float *d_origins, float *d_directions, float *d_viewdirs,
                           float *d_radii, float *d_imageplane, void *stream);

/* Stage entry points (same device code as the fused kernel; used by the
 * parity tests and for drop-in use of the individual reference functions). */

/* stepfun.sample_intervals (stepfun.py:209-258) with deterministic centres.
 * d_t [R,M+1], d_logits [R,M] -> d_sdist [R,N+1], d_bin_idx [R,N] (optional). */
int refnerf_sample_intervals(const float *d_t, const float *d_logits, int32_t R, int32_t M,
                             int32_t N, float s_min, float s_max, float *d_sdist,
                             int32_t *d_bin_idx, void *stream);

/* coord.integrated_pos_enc(min_deg=0,max_deg=16) on lifted Gaussians
 * (coord.py:107-126): d_lmean/d_lvar [n,3] -> d_feat [n,96]. */
int refnerf_integrated_pos_enc(const float *d_lmean, const float *d_lvar, int32_t n,
                               float *d_feat, void *stream);

/* ref_utils.generate_ide_fn(5) (ref_utils.py:98-161): d_xyz [n,3],
 * d_kappa_inv [n] -> d_ide [n,72]. */
int refnerf_integrated_dir_enc(const float *d_xyz, const float *d_kappa_inv, int32_t n,
                               float *d_ide, void *stream);

/* render.compute_alpha_weights (render.py:132-149) + render.volumetric_rendering (render.py:152-254, all five
 * render-time `srgb_mapping` modes :186-216, float64 percentiles via stepfun.py:294-307 / math.py:114-142) on
 * caller-supplied per-sample values -- the compositing phase of the fused level kernel behind its own entry.
 * d_density / d_roughness [R,N]; d_tdist [R,N+1]; d_directions [R,3]; d_far [R]; the [R,N,3] inputs may be NULL
 * (zero).  Reads cfg->n_samples (<= 1024), opaque_background, render_srgb_mode, compute_extras, bg_rgb; writes
 * out->d_weights and the d_r_* fields (d_r_normals only when d_normals is given). */
int refnerf_render_rays(const refnerf_level_cfg *cfg, int32_t R, const float *d_density, const float *d_tdist,
                        const float *d_directions, const float *d_far, const float *d_rgb, const float *d_diffuse,
                        const float *d_specular, const float *d_normals, const float *d_normals_pred,
                        const float *d_roughness, const float *d_tint, const refnerf_level_out *out, void *stream);

/* The three Ref-NeRF losses of one level, fused (internal/train_utils.py:33-88 compute_data_loss with
 * data_loss_type 'mse', :165-183 orientation_loss, :186-204 predicted_normal_loss): ONE pass over the level's outputs
 * instead of ~25 elementwise ATen ops and their autograd graph.
 * refnerf_losses_forward writes per-ray terms d_terms [R,3] = { sum_c lossmult (rgb_c - gt_c)^2,
 * sum_i w_i min(0, n_i . (-viewdir))^2, sum_i w_i (1 - n_i . n_pred,i) }; the caller sums them over the rays and applies
 * the normalisers (sum of the broadcast lossmult; ray count) and Config's coarse / fine multipliers.
 * d_orientation_normals = ray_history[Config.orientation_loss_target] (NULL: term off); d_normals = the density normals
 * (NULL: predicted-normal term off).  refnerf_losses_backward produces what autograd would hand to the level's
 * backward: dL/d renderings['rgb'] [R,3], dL/d ray_history['weights'] [R,N], dL/d ray_history['normals_pred'] [R,N,3],
 * given g_* = d(term)/d(term sum) (multiplier / normaliser included) times d_upstream[0..2] = dL/d(data, orientation,
 * predicted-normal term) (device float[3], NULL = 1: on the device so that autograd's grad_output needs no host sync); the density normals are detached as in the reference
 * (models.py:609), so an orientation target other than normals_pred only reaches the weights. */
int refnerf_losses_forward(int32_t R, int32_t N, const float *d_r_rgb, const float *d_gt_rgb, const float *d_lossmult,
                           const float *d_weights, const float *d_orientation_normals, const float *d_normals,
                           const float *d_normals_pred, const float *d_viewdirs, float *d_terms, void *stream);
int refnerf_losses_backward(int32_t R, int32_t N, const float *d_r_rgb, const float *d_gt_rgb, const float *d_lossmult,
                            const float *d_weights, const float *d_orientation_normals, int32_t orientation_on_pred,
                            const float *d_normals, const float *d_normals_pred, const float *d_viewdirs,
                            float g_data, float g_orientation, float g_normal, const float *d_upstream,
                            float *d_g_r_rgb, float *d_g_weights, float *d_g_normals_pred, void *stream);

/* Total duration (ms) and count of the `refnerf_level_forward` kernels launched
 * since refnerf_set_timing(1), from HIP event pairs on the launch stream.
 * Used by bench.py for the roofline line. */
enum { REFNERF_TIMER_FORWARD = 0,   /* the level kernel of refnerf_level_forward / _forward_train           */
       REFNERF_TIMER_BACKWARD = 1,  /* the per-sample backward kernel of refnerf_level_backward              */
       REFNERF_TIMER_WGRAD = 2 };   /* its weight-gradient GEMM                                              */
int refnerf_set_timing(int enable);
int refnerf_get_timing(double *total_ms, int64_t *launches);                         /* REFNERF_TIMER_FORWARD */
int refnerf_get_timing_family(int family, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
