/*
 * refnerf_device_math.h -- per-sample device math of the Ref-NeRF path.
 *
 * Every function restates the cited reference lines in the reference's fp32
 * operation order (build with -ffp-contract=off: no implicit fma), so that the
 * sample means that feed the (chaotic, up to 2^15 x) IPE sines are bit-equal
 * to the reference CPU path.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "refnerf_detmath.h"
#include "refnerf_layout.h"
#include "refnerf_ide_tables.h"

namespace rn {

constexpr float EPS32 = 1.1920928955078125e-07f; /* torch.finfo(float32).eps */
constexpr float HALF_PI_F = 1.57079637050628662109375f;   /* fl32(0.5*pi) */
constexpr float T100PI = 314.159271240234375f;            /* fl32(100*pi) */
constexpr float LOG3_F = 1.09861228466033935546875f;      /* fl32(log 3)  */
constexpr float LOG2E_F = 1.44269504088896341f;
constexpr float LN2_F = 0.69314718055994531f;
constexpr float INV_2PI_F = 0.15915494309189535f;


__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* coord.py:96-98 (fn=None) */
__device__ __forceinline__ float s_to_t(float s, float nearv, float farv) {
  float a = s * farv;
  float b = (1.0f - s) * nearv;
  return a + b;
}

/* coord.construct_ray_warps for the other ray-distance functions (coord.py:63-99): fn_inv(s fn(far) + (1 - s) fn(near)) */
__device__ __forceinline__ float raydist_fwd(float x, int m) {
  switch (m) {
    case REFNERF_RAYDIST_PIECEWISE: return x < 1.0f ? 0.5f * x : 1.0f - 0.5f / x;
    case REFNERF_RAYDIST_RECIPROCAL: return 1.0f / x;
    case REFNERF_RAYDIST_LOG: return logf(x);
    case REFNERF_RAYDIST_EXP: return expf(x);
    case REFNERF_RAYDIST_SQRT: return sqrtf(x);
    case REFNERF_RAYDIST_SQUARE: return x * x;
    default: return x;
  }
}
__device__ __forceinline__ float raydist_inv(float x, int m) {
  switch (m) {
    case REFNERF_RAYDIST_PIECEWISE: return x < 0.5f ? 2.0f * x : 0.5f / (1.0f - x);
    case REFNERF_RAYDIST_RECIPROCAL: return 1.0f / x;
    case REFNERF_RAYDIST_LOG: return expf(x);
    case REFNERF_RAYDIST_EXP: return logf(x);
    case REFNERF_RAYDIST_SQRT: return x * x;
    case REFNERF_RAYDIST_SQUARE: return sqrtf(x);
    default: return x;
  }
}
__device__ __forceinline__ float s_to_t(float s, float nearv, float farv, int raydist) {
  if (raydist == REFNERF_RAYDIST_NONE) return s_to_t(s, nearv, farv);
  const float sn = raydist_fwd(nearv, raydist), sf = raydist_fwd(farv, raydist);
  const float a = s * sf;
  const float b = (1.0f - s) * sn;
  return raydist_inv(a + b, raydist);
}

/* torch.linspace(pad, 1-pad-eps, N)[k] (stepfun.py:199-204; ATen fills
 * symmetrically with fused multiply-adds). */
__device__ __forceinline__ float linspace_u(int k, int n) {
  double pad = 1.0 / (2.0 * (double)n);
  float start = (float)pad;
  float end = (float)(1.0 - pad - (double)EPS32);
  float step = (end - start) / (float)(n - 1);
  return (k < n / 2) ? fmaf(step, (float)k, start) : fmaf(-step, (float)(n - 1 - k), end);
}

/* render.py:46-80 (conical frustum) / 83-102 (cylinder): moments of the interval [t0, t1] along the ray */
__device__ __forceinline__ void frustum_moments(float radius, float t0, float t1, int ray_shape, float &t_mean, float &t_var, float &r_var) {
  if (ray_shape == 0) {
    float mu = (t0 + t1) / 2.0f;
    float hw = (t1 - t0) / 2.0f;
    float hw2 = hw * hw, mu2 = mu * mu;
    float den = fmaxf(EPS32, 3.0f * mu2 + hw2);
    t_mean = mu + ((2.0f * mu) * hw2) / den;
    float hw4 = (float)((double)hw * (double)hw * (double)hw * (double)hw);
    const float c415 = (float)(4.0 / 15.0);
    t_var = hw2 / 3.0f - ((c415 * hw4) * (12.0f * mu2 - hw2)) / (den * den);
    r_var = (mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (c415 * hw4) / den;
    r_var = r_var * (radius * radius);
  } else {
    t_mean = (t0 + t1) / 2.0f;
    r_var = (radius * radius) / 4.0f;
    float dt = t1 - t0;
    t_var = (dt * dt) / 12.0f;
  }
}

/* render.py:46-80 / 83-102 + 22-43 + coord.py:129-133 (octahedron/1 basis):
 * lifted mean (-z,-y,-x) and lifted variance (C_zz,C_yy,C_xx). */
__device__ __forceinline__ void cast_sample(const float o[3], const float d[3], float radius, float t0,
                                            float t1, int ray_shape, float lmean[3], float lvar[3]) {
  float t_mean, t_var, r_var;
  frustum_moments(radius, t0, t1, ray_shape, t_mean, t_var, r_var);
  float dms = fmaxf(1e-10f, (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  float mean[3], cd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mean[i] = d[i] * t_mean + o[i];
    float d_outer = d[i] * d[i];
    float null_outer = 1.0f - d[i] * (d[i] / dms);
    cd[i] = t_var * d_outer + r_var * null_outer;
  }
  lmean[0] = -mean[2]; lmean[1] = -mean[1]; lmean[2] = -mean[0];
  lvar[0] = cd[2]; lvar[1] = cd[1]; lvar[2] = cd[0];
}

/* The same for a general basis (NerfMLP.basis_shape / basis_subdivisions, geopoly.generate_basis): the Gaussian with its
 * FULL covariance (render.py:22-43 with diag = False, as models.py:214-227 calls it), then coord.lift_and_diagonalize
 * (coord.py:129-133) onto three directions b[0..2] (rows of the basis; the image keeps them in the reference's component
 * order): lifted mean = mean . b (fma chain over x, y, z: the order of ATen's [.., 3] x [3, n] product at batch sizes
 * that matter), lifted variance = sum_i b_i (sum_k C_ik b_k). */
__device__ __forceinline__ void cast_sample_full(const float o[3], const float d[3], float radius, float t0, float t1, int ray_shape,
                                                 float mean[3], float cov[9]) {
  float t_mean, t_var, r_var;
  frustum_moments(radius, t0, t1, ray_shape, t_mean, t_var, r_var);
  const float dms = fmaxf(1e-10f, (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mean[i] = d[i] * t_mean + o[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float d_outer = d[i] * d[k];
      const float null_outer = (i == k ? 1.0f : 0.0f) - d[i] * (d[k] / dms);
      cov[3 * i + k] = t_var * d_outer + r_var * null_outer;
    }
  }
}
__device__ __forceinline__ void lift_onto(const float mean[3], const float cov[9], const float b[3], float &lm, float &lv) {
  lm = fmaf(mean[2], b[2], fmaf(mean[1], b[1], mean[0] * b[0]));
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float cb = fmaf(cov[3 * i + 2], b[2], fmaf(cov[3 * i + 1], b[1], cov[3 * i] * b[0]));
    acc = (i == 0) ? b[0] * cb : acc + b[i] * cb;
  }
  lv = acc;
}

/* math.py:22-34: where(|x| < 100pi, x, x % 100pi), floored remainder */
__device__ __forceinline__ float safe_arg(float x) {
  if (fabsf(x) < T100PI) return x;
  float m = fmodf(x, T100PI);
  if (m != 0.0f && m < 0.0f) m += T100PI;
  return m;
}

/* a = k pi/2 + r, |r| <= pi/4 (+ 1e-5), for |a| < 400: k = rint(a 2/pi) in fp32 and a three-constant Cody-Waite subtraction --
 * k < 2^8, pi/2 = 1.5703125 (8 bits) + 4.8375e-4 (11 bits) + 7.5498e-8: the first two products are exact, the reduction's error
 * is the last FMA's rounding (6e-8 r) + 2e-15 k.  Rounds 3-4 did this in float64 (rint + fma, two half-rate instructions plus
 * three conversions per feature): the 96 IPE features were 15 k of a spatial run's 250 k cycles; in fp32 the same kernels measure
 * the same 9.2e-8 maximum error against float64 over 4e6 IPE arguments, 6 results per million differ from the float64 reduction
 * by one ulp (where k itself differs: r a hair past pi/4, inside the kernels' range). */
__device__ __forceinline__ void quadrant_reduce(float a, float &r, int &q) {
  const float k = __builtin_rintf(a * 0.636619772367581343f);
  r = fmaf(-k, 7.54978995489188216e-8f, fmaf(-k, 4.837512969970703125e-4f, fmaf(-k, 1.5703125f, a)));
  q = (int)k;
}
/* sin(a) for |a| < 400 (safe_arg's range is |a| < 100 pi): 4-term minimax kernels in fp32; <= 9.2e-8 absolute (1.5 ulp) over
 * |a| < 315 (checked against float64).  The library's sinf carries its large-argument (Payne-Hanek) path through every call:
 * ~150 instructions against ~20. */
__device__ __forceinline__ float sin_reduced(float a) {
  float r; int q;
  quadrant_reduce(a, r, q);
  const float r2 = r * r;
  const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
  const float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
/* cos(a), same reduction and kernels */
__device__ __forceinline__ float cos_reduced(float a) {
  float r; int q;
  quadrant_reduce(a, r, q);
  const float r2 = r * r;
  const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
  const float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const float v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}
/* safe_arg (math.py:22-34: where(|x| < 100 pi, x, x % 100 pi), torch's floored remainder) without branches, the library's fmodf
 * loop (which hipcc inlines as a divergent 12-bits-per-trip reduction) or float64: q = floor(x / T + 3e-4) is the true floor or
 * one more (x / T in fp32 is off by < 1.5e-4 for |x| < 2^19), so r = fma(-q, T, x) lies in [-0.1, T) and is EXACT -- x and T are
 * multiples of 2^-15 there, |r| < 512 -- and a negative r takes + T, exactly, as torch's `%` does.  Equal to the float64
 * evaluation on 4e6 IPE arguments (|x| up to 4.4e5), bit for bit. */
__device__ __forceinline__ float safe_arg_exact(float x) {
  const float q = __builtin_floorf(fmaf(x, 1.0f / T100PI, 3e-4f));
  float r = fmaf(-q, T100PI, x);
  r = (r < 0.0f) ? r + T100PI : r;
  return (fabsf(x) < T100PI) ? x : r;
}
/* IPE feature of the split-f16 kernel: the reference's argument (fp32 product, fp32 + pi/2, the fp32 `mod 100 pi`), then
 * sin_reduced and the hardware exp2 -- absolute error < 2e-7, below the 2^-22 of the hi + lo split it feeds */
__device__ __forceinline__ float ipe_feature_split(float lm, float lv, int j, int cos_block) {
  const float sc = __builtin_ldexpf(1.0f, j), sc2 = __builtin_ldexpf(1.0f, 2 * j);
  float x = lm * sc;
  if (cos_block) x = x + HALF_PI_F;
  const float e = __builtin_amdgcn_exp2f((-0.5f * LOG2E_F) * (lv * sc2));
  return e * sin_reduced(safe_arg_exact(x));
}

/* One IPE feature (coord.py:119-126): block 0 = sin, block 1 = "cos" =
 * sin(fl(x + pi/2)). */
template <bool FAST = false>
__device__ __forceinline__ float ipe_feature(float lm, float lv, int j, int cos_block) {
  float sc = __builtin_ldexpf(1.0f, j), sc2 = __builtin_ldexpf(1.0f, 2 * j);
  if (FAST) {
    /* sin is 2pi-periodic, so safe_sin's "mod 100pi" is the identity up to
     * rounding: reduce in revolutions and use v_sin_f32 (input in revolutions) */
    float r = (lm * sc) * INV_2PI_F;
    if (cos_block) r = r + 0.25f;
    r = __builtin_amdgcn_fractf(r);
    float e = __builtin_amdgcn_exp2f((-0.5f * LOG2E_F) * (lv * sc2));
    return e * __builtin_amdgcn_sinf(r);
  }
  float x = lm * sc;
  if (cos_block) x = x + HALF_PI_F;
  float e = expf(-0.5f * (lv * sc2));
  return e * sinf(safe_arg(x));
}

/* Transcendentals.  FAST = false: the accurate ocml routines in the reference's
 * operation order (fp32 parity mode).  FAST = true (bf16 mode, whose MLP inputs
 * are rounded to 8 bits anyway): hardware v_exp/v_log/v_rcp/v_sin. */
template <bool FAST> __device__ __forceinline__ float m_exp(float x) {
  if (FAST) return __builtin_amdgcn_exp2f(x * LOG2E_F);
  return expf(x);
}
template <bool FAST> __device__ __forceinline__ float m_log(float x) {
  if (FAST) return __builtin_amdgcn_logf(x) * LN2_F;
  return logf(x);
}
template <bool FAST> __device__ __forceinline__ float m_div(float a, float b) {
  if (FAST) return a * __builtin_amdgcn_rcpf(b);
  return a / b;
}
template <bool FAST> __device__ __forceinline__ float softplus_m(float x) {
  if (FAST) return x > 20.0f ? x : m_log<true>(1.0f + m_exp<true>(x));
  return x > 20.0f ? x : log1pf(expf(x));
}
template <bool FAST> __device__ __forceinline__ float sigmoid_m(float x) { return m_div<FAST>(1.0f, 1.0f + m_exp<FAST>(-x)); }
__device__ __forceinline__ float softplus_t(float x) { return softplus_m<false>(x); }
__device__ __forceinline__ float sigmoid_t(float x) { return sigmoid_m<false>(x); }

/* image.py:51-59 */
template <bool FAST = false>
__device__ __forceinline__ float linear_to_srgb(float x) {
  float srgb0 = (float)(323.0 / 25.0) * x;
  float pw = FAST ? __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fmaxf(EPS32, x)) * (float)(5.0 / 12.0))
                  : powf(fmaxf(EPS32, x), (float)(5.0 / 12.0));
  float srgb1 = FAST ? (211.0f * pw - 11.0f) * (1.0f / 200.0f) : (211.0f * pw - 11.0f) / 200.0f;
  return (x <= 0.0031308f) ? srgb0 : srgb1;
}

/* ref_utils.py:98-161, deg_view = 5, evaluated with the stable normalised
 * Legendre recurrence (same polynomials as the reference's Vandermonde form;
 * SURVEY.md H2).  part = 0: real parts (out[0..35]), 1: imaginary parts.
 * `emit(q, value)` receives term index q in the reference's (l,m) order. */
template <bool FAST = false, typename Emit>
__device__ __forceinline__ void ide_eval(float x, float y, float z, float kappa_inv, int part, Emit emit) {
  float att1 = m_exp<FAST>(-1.0f * kappa_inv), att2 = m_exp<FAST>(-3.0f * kappa_inv), att4 = m_exp<FAST>(-10.0f * kappa_inv);
  float att8 = m_exp<FAST>(-36.0f * kappa_inv), att16 = m_exp<FAST>(-136.0f * kappa_inv);
  float pr = 1.0f, pi = 0.0f;
#pragma unroll
  for (int m = 0; m <= 16; ++m) {
    if (m > 0) { float nr = pr * x - pi * y; float ni = pr * y + pi * x; pr = nr; pi = ni; }
    const float pw = part ? pi : pr;
    float tm2 = 0.0f, tm1 = IDE_C[m];
#pragma unroll
    for (int l = m; l <= 16; ++l) {
      float tl;
      if (l == m) tl = IDE_C[m];
      else { tl = IDE_A[m][l] * (z * tm1 - IDE_B[m][l] * tm2); tm2 = tm1; tm1 = tl; }
      if (l == 1) emit(0 + m, pw * (tl * att1));
      if (l == 2) emit(2 + m, pw * (tl * att2));
      if (l == 4) emit(5 + m, pw * (tl * att4));
      if (l == 8) emit(10 + m, pw * (tl * att8));
      if (l == 16) emit(19 + m, pw * (tl * att16));
    }
  }
}

/* coord.pos_enc(d, 0, 5, append_identity = True) (coord.py:136-147; MLP.use_directional_enc = False, models.py:487-492) with
 * the calling convention of ide_eval: part 0 emits slots 0..35 = [x y z | sin(2^j d_i), j-major (15) | 0 x18], part 1 slots
 * 0..35 of the second half = [sin(2^j d_i + pi/2) (15) | 0 x21] (as the reference: a sine of the shifted argument). */
template <bool FAST = false, bool REDUCED = false, typename Emit>
__device__ __forceinline__ void posenc_eval(float x, float y, float z, int part, Emit emit) {
  const float d[3] = {x, y, z};
  const float half_pi = (float)(0.5 * 3.14159265358979323846);
#pragma unroll
  for (int q = 0; q < IDE_TERMS; ++q) {
    float v = 0.0f;
    if (q < 3 && part == 0) v = d[q];
    const int k = part ? q : q - 3;                        /* 3 j + i inside the sin / shifted-sin block */
    if (k >= 0 && k < 15) {
      const float sx = d[k % 3] * (float)(1 << (k / 3)) + (part ? half_pi : 0.0f);
      /* REDUCED (split-f16 kernel): |sx| < 34, sin_reduced is 1.5 ulp there at a sixth of the library routine's code */
      v = FAST ? __builtin_amdgcn_sinf(sx * (float)(0.5 / 3.14159265358979323846)) : (REDUCED ? sin_reduced(sx) : sinf(sx));
    }
    emit(q, v);
  }
}
/* its gradient w.r.t. d; g(q) = upstream gradient of slot q (0..35 first half, 36..71 second half) */
template <typename G>
__device__ __forceinline__ void posenc_grad(float x, float y, float z, G g, float (&gxyz)[3]) {
  const float d[3] = {x, y, z};
  const float half_pi = (float)(0.5 * 3.14159265358979323846);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float dep = 0.0f;
    float acc = g(i, dep);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float sc = (float)(1 << j), sx = d[i] * sc;
      /* (|sx| <= 16 |d|: the reduced kernels, 1.5 ulp -- the library's cosf carries its Payne-Hanek path and 16 registers of it
       *  into every caller) */
      acc += sc * (cos_reduced(sx) * g(3 + 3 * j + i, dep) + cos_reduced(sx + half_pi) * g(IDE_TERMS + 3 * j + i, dep));
    }
    gxyz[i] = acc;
  }
}

/* ---- derivatives used by the backward kernel ---- */
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)); }
/* d linear_to_srgb / du (image.py:51-59) */
__device__ __forceinline__ float srgb_grad(float u) {
  if (u <= 0.0031308f) return (float)(323.0 / 25.0);
  if (!(u > EPS32)) return 0.0f;
  return (211.0f / 200.0f) * (float)(5.0 / 12.0) * powf(u, (float)(5.0 / 12.0) - 1.0f);
}

/* Backward of a 3-channel "optionally normalise by max(max_c, 1), optionally
 * sRGB-encode, clip to [0,1]" map (models.py:712-723 and render.py:186-216),
 * with torch's subgradient conventions (clip passes on the closed interval,
 * maximum splits ties 1/2, amax shares among ties).  pre[]: inputs of the map;
 * g[]: upstream gradient in, gradient w.r.t. pre out. */
__device__ __forceinline__ void colour_map_backward(const float pre[3], bool normed, bool srgb, float g[3]) {
  float norm = 1.0f;
  const float mxc = fmaxf(fmaxf(pre[0], pre[1]), pre[2]);
  if (normed) norm = fmaxf(mxc, 1.0f);
  float gu[3], gnorm = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float u = pre[c] / norm;
    float yv = srgb ? linear_to_srgb(u) : u;
    float pass = (yv >= 0.0f && yv <= 1.0f) ? 1.0f : 0.0f;
    float dy = srgb ? srgb_grad(u) : 1.0f;
    gu[c] = g[c] * pass * dy;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { g[c] = gu[c] / norm; gnorm += -gu[c] * pre[c] / (norm * norm); }
  if (normed) {
    float gm = (mxc > 1.0f) ? gnorm : (mxc == 1.0f ? 0.5f * gnorm : 0.0f);
    if (gm != 0.0f) {
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) cnt += (pre[c] == mxc);
#pragma unroll
      for (int c = 0; c < 3; ++c) if (pre[c] == mxc) g[c] += gm / (float)cnt;
    }
  }
}

/* Gradient of ide_eval (stable form) w.r.t. (x,y,z) and kappa_inv.
 * g(q, dep) = upstream gradient of output q (0..35 real parts, 36..71 imaginary); `dep` is the recurrence value the read is
 * needed for: a caller that wants its 72 reads issued where they are used (not hoisted to the front and parked in scratch) ties
 * the read's address to it (asm volatile("" : "+v"(addr), "+v"(dep))). */
template <typename G>
__device__ __forceinline__ void ide_grad(float x, float y, float z, float kappa_inv, G g, float (&gxyz)[3], float &gkappa) {
  float att1 = expf(-1.0f * kappa_inv), att2 = expf(-3.0f * kappa_inv), att4 = expf(-10.0f * kappa_inv);
  float att8 = expf(-36.0f * kappa_inv), att16 = expf(-136.0f * kappa_inv);
  /* (pinned here: sunk to their first uses, the five exponentials make every product in front of them wait -- in scratch) */
  asm volatile("" : "+v"(att1), "+v"(att2), "+v"(att4), "+v"(att8), "+v"(att16));
  float pr = 1.0f, pi = 0.0f, prm1 = 0.0f, pim1 = 0.0f;
  float gx = 0.0f, gy = 0.0f, gz = 0.0f, gk = 0.0f;
#pragma unroll
  for (int m = 0; m <= 16; ++m) {
    if (m > 0) { prm1 = pr; pim1 = pi; pr = prm1 * x - pim1 * y; pi = prm1 * y + pim1 * x; }
    float gpr = 0.0f, gpi = 0.0f;
    float tm2 = 0.0f, tm1 = IDE_C[m], dm2 = 0.0f, dm1 = 0.0f;
#pragma unroll
    for (int l = m; l <= 16; ++l) {
      float tl, dl;
      if (l == m) { tl = IDE_C[m]; dl = 0.0f; }
      else {
        tl = IDE_A[m][l] * (z * tm1 - IDE_B[m][l] * tm2);
        dl = IDE_A[m][l] * (tm1 + z * dm1 - IDE_B[m][l] * dm2);
        /* (pinned: program order, one degree at a time.  Pure arithmetic is not ordered by sched_barrier -- left to itself the
         * instruction selector runs the whole T recurrence of an order m first and parks every T_l for the derivative recurrence
         * behind it: 77 scratch stores / 92 loads per sample in level_bwd_sq, round 6) */
        asm volatile("" : "+v"(tl), "+v"(dl));
        tm2 = tm1; tm1 = tl; dm2 = dm1; dm1 = dl;
      }
      if (l == 1 || l == 2 || l == 4 || l == 8 || l == 16) {
        const int idx = (l == 1 ? 0 : l == 2 ? 2 : l == 4 ? 5 : l == 8 ? 10 : 19) + m;
        const float sigma = (float)(0.5 * l * (l + 1));
        const float att = (l == 1 ? att1 : l == 2 ? att2 : l == 4 ? att4 : l == 8 ? att8 : att16);
        const float gre = g(idx, tl), gim = g(IDE_TERMS + idx, tl);
        const float s = tl * att;
        const float gs = gre * pr + gim * pi;
        gpr += gre * s; gpi += gim * s;
        gz += gs * att * dl;
        gk += gs * tl * (-sigma * att);
        asm volatile("" : "+v"(gz), "+v"(gk), "+v"(gpr), "+v"(gpi));      /* accumulate now */
      }
    }
    if (m > 0) {
      gx += (float)m * (gpr * prm1 + gpi * pim1);
      gy += (float)m * (-gpr * pim1 + gpi * prm1);
    }
  }
  gxyz[0] = gx; gxyz[1] = gy; gxyz[2] = gz;
  gkappa = gk;
}

}  // namespace rn
