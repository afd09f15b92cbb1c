/*
 * refnerf_oracle.c -- CPU restatement of the Ref-NeRF rendering inner loop.
 *
 * TEST INFRASTRUCTURE ONLY (see refnerf_oracle.h).  Parity status: PINNED by
 * the .npz files under tests/golden (captured from the upstream reference in the build
 * container by tests/golden/make_golden.py).
 *
 * Reference citations are file:line relative to the upstream repo root.
 * fp32 IEEE arithmetic in the reference's operation order; compile with
 *   gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif
#include "../include/refnerf_detmath.h"
typedef float rn_f32;   /* stays fp32 in the float64 build: the reference's fp32-derived constants (linspace, 100 pi, pi/2) */
/* -DRN_ORACLE_F64 (oracle/Makefile: librefnerf_oracle_f64.so): the SAME source with every fp32 quantity and libm
 * call switched to float64 -- the "truth" build that gates fp32 rounding error (of the reference's arithmetic and
 * of the HIP kernels alike) in tests/.  The fp32 build is untouched by it. */
#ifdef RN_ORACLE_F64
#include "refnerf_oracle_f64.h"
#endif
#include "refnerf_oracle.h"

#define EPS32 1.1920928955078125e-07f /* torch.finfo(float32).eps */

/* ------------------------------------------------------------------ */
/* parameter blob layout = state_dict order (models.py:497-531)       */
/* ------------------------------------------------------------------ */
void rn_param_layout(rn_param_offsets *o) {
  int p = 0;
  for (int i = 0; i < RN_DEPTH; ++i) {
    int in = (i == 0) ? RN_IPE_DIM
                      : ((i - 1) % RN_SKIP == 0 && (i - 1) > 0 ? RN_WIDTH + RN_IPE_DIM
                                                               : RN_WIDTH);
    o->sp_in[i] = in;
    o->sp_w[i] = p; p += RN_WIDTH * in;
    o->sp_b[i] = p; p += RN_WIDTH;
  }
  o->density_w = p; p += RN_WIDTH;      o->density_b = p; p += 1;
  o->gradpred_w = p; p += 3 * RN_WIDTH; o->gradpred_b = p; p += 3;
  o->rough_w = p; p += RN_WIDTH;        o->rough_b = p; p += 1;
  o->diffuse_w = p; p += 3 * RN_WIDTH;  o->diffuse_b = p; p += 3;
  o->tint_w = p; p += 3 * RN_WIDTH;     o->tint_b = p; p += 3;
  o->bneck_w = p; p += RN_BNECK * RN_WIDTH; o->bneck_b = p; p += RN_BNECK;
  for (int i = 0; i < RN_DEPTH; ++i) {
    int in = (i == 0) ? RN_DIR_IN
                      : ((i - 1) % RN_SKIP == 0 && (i - 1) > 0 ? RN_WIDTH + RN_DIR_IN
                                                               : RN_WIDTH);
    o->vd_in[i] = in;
    o->vd_w[i] = p; p += RN_WIDTH * in;
    o->vd_b[i] = p; p += RN_WIDTH;
  }
  o->rgb_w = p; p += 3 * RN_WIDTH; o->rgb_b = p; p += 3;
  o->total = p;
}

void rn_level_cfg_default(rn_level_cfg *c) {
  memset(c, 0, sizeof(*c));
  c->n_samples = 128; c->n_in = 1;
  c->training = 0; c->compute_extras = 1;
  c->srgb_mapping = 1; c->srgb_mapping_normalization = 1;
  c->render_srgb_mode = RN_SRGB_NONE;
  c->opaque_background = 0; c->ray_shape = 0; c->ide_mode = 0; c->raydist = 0; c->disable_integration = 0;
  c->anneal = 1.0f; c->resample_padding = 0.01f;
  c->s_near = 0.0f; c->s_far = 1.0f;
  c->density_bias = 0.5f; c->roughness_bias = -1.0f;
  c->rgb_premultiplier = 1.0f; c->rgb_bias = 0.0f; c->rgb_padding = 0.001f;
  c->bg_rgb = 1.0f;
}

/* ------------------------------------------------------------------ */
/* sampler                                                            */
/* ------------------------------------------------------------------ */

/* torch.linspace(start, end, n) for float32 (ATen RangeFactories: step in
 * fp32, symmetric fill, fused multiply-add) as used by stepfun.py:199-204:
 *   pad = 1/(2N); u = linspace(pad, 1 - pad - eps, N). */
void rn_linspace_u(int n, float *u) {
  double pad = 1.0 / (2.0 * n);
  rn_f32 start = (rn_f32)pad;
  rn_f32 end = (rn_f32)(1.0 - pad - (double)EPS32);
  if (n == 1) { u[0] = start; return; }
  rn_f32 step = (end - start) / (rn_f32)(n - 1);
  int half = n / 2;
  for (int i = 0; i < n; ++i)     /* fp32 in both builds: the quantiles are constants of the reference */
    u[i] = (i < half) ? __builtin_fmaf(step, (rn_f32)i, start)
                      : __builtin_fmaf(-step, (rn_f32)(n - 1 - i), end);
}

/* models.py:200-203 */
void rn_resample_logits(const float *t, const float *w, int M, float anneal,
                        float padding, float *logits) {
  for (int i = 0; i < M; ++i)
    logits[i] = (t[i + 1] > t[i]) ? anneal * rn_det_logf(w[i] + padding) : -INFINITY;   /* shared with the kernels */
}

static float nan_to_num0(float x) { /* torch.nan_to_num(x, 0) */
  if (isnan(x)) return 0.0f;
  if (isinf(x)) return x > 0 ? FLT_MAX : -FLT_MAX;
  return x;
}
static float clip01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* stepfun.py:209-258 -> sample (168-206) -> invert_cdf (157-165)
 * -> integrate_weights (134-154) -> math.sorted_interp (math.py:88-111). */
void rn_sample_intervals(const float *t, const float *w_logits, int M, int N,
                         float smin, float smax, float *sdist, int32_t *bin_idx) {
  float *p = (float *)malloc(sizeof(float) * (size_t)(M + (M + 1) + N + N));
  float *cw = p + M, *u = cw + (M + 1), *c = u + N;
  /* softmax (stepfun.py:160); exp through the shared bit-reproducible
   * rn_det_expf so that the HIP path can match the CDF bit for bit. */
  float mx = -INFINITY;
  for (int i = 0; i < M; ++i) mx = fmaxf(mx, w_logits[i]);
  float sum = 0.0f;
  for (int i = 0; i < M; ++i) { p[i] = rn_det_expf(w_logits[i] - mx); sum += p[i]; }
  for (int i = 0; i < M; ++i) p[i] = p[i] / sum;
  /* integrate_weights: cw = [0, min(1, cumsum(w[:-1])), 1]; torch.cumsum on
   * CPU accumulates float in double (at::acc_type<float,false>). */
  cw[0] = 0.0f;
  double acc = 0.0;
  for (int i = 0; i < M - 1; ++i) {
    acc += (double)p[i];
    cw[i + 1] = fminf(1.0f, (float)acc);
  }
  cw[M] = 1.0f;
  rn_linspace_u(N, u);
  /* sorted_interp: for each u find the bracketing CDF knots by value. */
  for (int k = 0; k < N; ++k) {
    float x = u[k];
    int lo = -1;
    for (int j = 0; j <= M; ++j) if (x >= cw[j]) lo = j;  /* mask is monotone */
    float xp0, fp0, xp1, fp1;
    /* max over masked entries, default element 0 */
    xp0 = cw[0]; fp0 = t[0];
    for (int j = 0; j <= M; ++j) if (x >= cw[j]) { xp0 = fmaxf(xp0, cw[j]); fp0 = fmaxf(fp0, t[j]); }
    xp1 = cw[M]; fp1 = t[M];
    for (int j = 0; j <= M; ++j) if (!(x >= cw[j])) { xp1 = fminf(xp1, cw[j]); fp1 = fminf(fp1, t[j]); }
    float off = clip01(nan_to_num0((x - xp0) / (xp1 - xp0)));
    c[k] = fp0 + off * (fp1 - fp0);
    if (bin_idx) bin_idx[k] = lo < 0 ? 0 : lo;
  }
  /* stepfun.py:247-257: midpoints + reflected, clamped end fenceposts */
  for (int k = 0; k < N - 1; ++k) sdist[k + 1] = (c[k + 1] + c[k]) / 2.0f;
  sdist[0] = fmaxf(smin, 2.0f * c[0] - sdist[1]);
  sdist[N] = fminf(smax, 2.0f * c[N - 1] - sdist[N - 1]);
  free(p);
}

/* coord.py:96-98 (fn=None): t = s*far + (1-s)*near */
float rn_s_to_t(float s, float near, float far) {
  float a = s * far;
  float b = (1.0f - s) * near;
  return a + b;
}

/* coord.construct_ray_warps (coord.py:63-99) for the ray-distance functions the reference accepts:
 * s_to_t(s) = fn_inv(s * fn(far) + (1 - s) * fn(near)) */
static float raydist_fwd(float x, int m) {
  switch (m) {
    case 1: return x < 1.0f ? 0.5f * x : 1.0f - 0.5f / x;
    case 2: return 1.0f / x;
    case 3: return logf(x);
    case 4: return expf(x);
    case 5: return sqrtf(x);
    case 6: return x * x;
    default: return x;
  }
}
static float raydist_inv(float x, int m) {
  switch (m) {
    case 1: return x < 0.5f ? 2.0f * x : 0.5f / (1.0f - x);
    case 2: return 1.0f / x;
    case 3: return expf(x);
    case 4: return logf(x);
    case 5: return x * x;
    case 6: return sqrtf(x);
    default: return x;
  }
}
float rn_s_to_t_fn(float s, float near, float far, int raydist) {
  if (raydist == 0) return rn_s_to_t(s, near, far);
  float sn = raydist_fwd(near, raydist), sf = raydist_fwd(far, raydist);
  float a = s * sf;
  float b = (1.0f - s) * sn;
  return raydist_inv(a + b, raydist);
}

/* ------------------------------------------------------------------ */
/* ray casting                                                        */
/* ------------------------------------------------------------------ */

/* render.py:105-129 (cast_rays) -> 46-80 (conical_frustum_to_gaussian,
 * stable) / 83-102 (cylinder) -> 22-43 (lift_gaussian, diag=False), then
 * coord.py:129-133 (lift_and_diagonalize) with the octahedron/1 basis
 * pos_basis_t = [[0,0,-1],[0,-1,0],[-1,0,0]] (geopoly.py:78-123), which makes
 * lifted mean = (-z,-y,-x) and lifted var = (C_zz, C_yy, C_xx) exactly. */
void rn_cast_sample(const float *o, const float *d, float radius, float t0,
                    float t1, int ray_shape, float *lmean, float *lvar,
                    float *mean_xyz) {
  float t_mean, t_var, r_var;
  if (ray_shape == 0) {
    float mu = (t0 + t1) / 2.0f;
    float hw = (t1 - t0) / 2.0f;
    float hw2 = hw * hw, mu2 = mu * mu;
    float den = fmaxf(EPS32, 3.0f * mu2 + hw2);
    t_mean = mu + ((2.0f * mu) * hw2) / den;
    /* hw**4 is torch.pow(x, 4) (Sleef powf, <=1 ulp); use the correctly
     * rounded value. */
    float hw4 = (float)((double)hw * (double)hw * (double)hw * (double)hw);
    float c415 = (float)(4.0 / 15.0);
    t_var = hw2 / 3.0f - ((c415 * hw4) * (12.0f * mu2 - hw2)) / (den * den);
    r_var = (mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (c415 * hw4) / den;
    r_var = r_var * (radius * radius);
  } else {
    t_mean = (t0 + t1) / 2.0f;
    r_var = (radius * radius) / 4.0f;
    float dt = t1 - t0;
    t_var = (dt * dt) / 12.0f;
  }
  float dms = fmaxf(1e-10f, (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  float mean[3], cov_diag[3];
  for (int i = 0; i < 3; ++i) {
    mean[i] = d[i] * t_mean + o[i];
    float d_outer = d[i] * d[i];
    float null_outer = 1.0f - d[i] * (d[i] / dms);
    cov_diag[i] = t_var * d_outer + r_var * null_outer;
  }
  lmean[0] = -mean[2]; lmean[1] = -mean[1]; lmean[2] = -mean[0];
  lvar[0] = cov_diag[2]; lvar[1] = cov_diag[1]; lvar[2] = cov_diag[0];
  if (mean_xyz) { mean_xyz[0] = mean[0]; mean_xyz[1] = mean[1]; mean_xyz[2] = mean[2]; }
}

/* math.py:22-34: sin(where(|x| < 100pi, x, x % 100pi)); torch `%` is the
 * floored remainder, with the scalar cast to fp32. */
static float safe_arg(float x) {
  const float T = (rn_f32)(100.0 * M_PI);       /* the scalar is cast to fp32 by torch (also in the float64 build) */
  if (fabsf(x) < T) return x;
  float m = fmodf(x, T);
  if (m != 0.0f && (m < 0.0f)) m += T;
  return m;
}

/* coord.py:107-126 (integrated_pos_enc, min_deg=0, max_deg=16) + 102-104. */
void rn_ipe(const float *lmean, const float *lvar, float *feat) {
  const float HALF_PI = (rn_f32)(0.5 * M_PI);
  for (int j = 0; j < RN_IPE_DEG; ++j) {
    float sc = ldexpf(1.0f, j), sc2 = ldexpf(1.0f, 2 * j);
    for (int b = 0; b < 3; ++b) {
      float x = lmean[b] * sc;
      float s = lvar[b] * sc2;
      float e = expf(-0.5f * s);
      feat[j * 3 + b] = e * sinf(safe_arg(x));
      feat[48 + j * 3 + b] = e * sinf(safe_arg(x + HALF_PI));
    }
  }
}

/* ------------------------------------------------------------------ */
/* integrated directional encoding                                    */
/* ------------------------------------------------------------------ */

static const int IDE_L[5] = {1, 2, 4, 8, 16};

static double fact(int n) { double r = 1; for (int i = 2; i <= n; ++i) r *= i; return r; }

/* ref_utils.py:53-81 */
static double sph_harm_coeff(int l, int m, int k) {
  double a = 0.5 * (l + k + m - 1.0);
  double gb = 1.0;
  for (int i = 0; i < l; ++i) gb *= (a - i);
  gb /= fact(l);
  double al = ((m & 1) ? -1.0 : 1.0) * pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) * gb;
  return sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * al;
}

/* ref_utils.py:98-161 evaluated in float64 (the "truth" used to gate both the
 * reference's and the build's fp32 rounding error). */
void rn_ide_f64(const double *xyz, double kappa_inv, double *out) {
  double x = xyz[0], y = xyz[1], z = xyz[2];
  double zp[17]; zp[0] = 1; for (int i = 1; i <= 16; ++i) zp[i] = zp[i - 1] * z;
  double pr[17], pi[17]; pr[0] = 1; pi[0] = 0;
  for (int m = 1; m <= 16; ++m) { pr[m] = pr[m - 1] * x - pi[m - 1] * y; pi[m] = pr[m - 1] * y + pi[m - 1] * x; }
  int idx = 0;
  for (int li = 0; li < 5; ++li) {
    int l = IDE_L[li];
    double att = exp(-0.5 * l * (l + 1) * kappa_inv);
    for (int m = 0; m <= l; ++m, ++idx) {
      double s = 0;
      for (int k = 0; k <= l - m; ++k) s += zp[k] * sph_harm_coeff(l, m, k);
      out[idx] = pr[m] * s * att;
      out[RN_IDE_TERMS + idx] = pi[m] * s * att;
    }
  }
}

/* Reference-order fp32 evaluation: monomial Vandermonde in z times the fp32
 * coefficient matrix (ref_utils.py:117-151), complex64 powers of (x+iy).
 * Used only to pin the oracle against the golden IDE vectors. */
void rn_ide_ref_f32(const float *xyz, float kappa_inv, float *out) {
  static float mat[17][RN_IDE_TERMS];
  static int init = 0;
  if (!init) {
    memset(mat, 0, sizeof(mat));
    int idx = 0;
    for (int li = 0; li < 5; ++li) for (int m = 0; m <= IDE_L[li]; ++m, ++idx)
      for (int k = 0; k <= IDE_L[li] - m; ++k) mat[k][idx] = (float)sph_harm_coeff(IDE_L[li], m, k);
    init = 1;
  }
  float x = xyz[0], y = xyz[1], z = xyz[2];
  float zp[17];
  for (int i = 0; i <= 16; ++i) zp[i] = (i == 0) ? 1.0f : (i == 1 ? z : (i == 2 ? z * z : (i == 3 ? z * z * z : powf(z, (float)i))));
  float pr[17], pi[17]; pr[0] = 1; pi[0] = 0;
  for (int m = 1; m <= 16; ++m) { pr[m] = pr[m - 1] * x - pi[m - 1] * y; pi[m] = pr[m - 1] * y + pi[m - 1] * x; }
  int idx = 0;
  for (int li = 0; li < 5; ++li) {
    int l = IDE_L[li];
    float sigma = (float)(0.5 * l * (l + 1));
    float att = expf(-sigma * kappa_inv);
    for (int m = 0; m <= l; ++m, ++idx) {
      float s = 0.0f;
      for (int k = 0; k <= 16; ++k) s = fmaf(zp[k], mat[k][idx], s);
      out[idx] = (pr[m] * s) * att;
      out[RN_IDE_TERMS + idx] = (pi[m] * s) * att;
    }
  }
}

/* Stable fp32 evaluation of the SAME polynomials (documented deviation from
 * the reference's evaluation ORDER, not from its function -- SURVEY.md H2):
 * T_l^m(z) = sum_k z^k mat[k,(l,m)] obeys the fully-normalised Legendre
 * three-term recurrence
 *   T_m^m = c_m,  T_l^m = a_lm * (z*T_{l-1}^m - b_lm*T_{l-2}^m)
 * with a_lm = sqrt((4l^2-1)/(l^2-m^2)), b_lm = sqrt(((l-1)^2-m^2)/(4(l-1)^2-1)).
 * Tables: tab_c[17], tab_a[17][17], tab_b[17][17] (double -> fp32). */
static float g_ide_c[17], g_ide_a[17][17], g_ide_b[17][17];
static int g_ide_init = 0;
void rn_ide_tables(float *c, float *a, float *b) {
  for (int m = 0; m <= 16; ++m) {
    /* c_m = (-1)^m sqrt((2m+1)/(4pi (2m)!)) (2m-1)!! */
    double df = 1; for (int i = 1; i <= m; ++i) df *= (2 * i - 1);
    double cm = ((m & 1) ? -1.0 : 1.0) * sqrt((2.0 * m + 1.0) / (4.0 * M_PI * fact(2 * m))) * df;
    c[m] = (float)cm;
    for (int l = 0; l <= 16; ++l) {
      a[m * 17 + l] = 0; b[m * 17 + l] = 0;
      if (l > m) {
        a[m * 17 + l] = (float)sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
        if (l > m + 1)
          b[m * 17 + l] = (float)sqrt((((double)l - 1) * (l - 1) - (double)m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
      }
    }
  }
}
/* coord.pos_enc(direction, min_deg = 0, max_deg = 5, append_identity = True) (internal/coord.py:136-147, the
 * `use_directional_enc = False` branch of internal/models.py:487-492) written into the 72 slots of the directional
 * encoding: [x y z | sin(2^j x_i), j-major (15) | 0 x18 || sin(2^j x_i + pi/2) (15) | 0 x21].  The weight columns of the 33
 * features are embedded at those slots (refnerf_pl_amd.layout.variant_layout); fewer degrees use a prefix of each block. */
#define RN_POSENC_DEG 5
void rn_posenc_slots_f32(const float *xyz, float *out) {
  for (int q = 0; q < RN_IDE_DIM; ++q) out[q] = 0.0f;
  const float half_pi = (rn_f32)(0.5 * 3.14159265358979323846);
  for (int i = 0; i < 3; ++i) out[i] = xyz[i];
  for (int j = 0; j < RN_POSENC_DEG; ++j)
    for (int i = 0; i < 3; ++i) {
      float sx = xyz[i] * (float)(1 << j);
      out[3 + 3 * j + i] = sinf(sx);
      out[RN_IDE_DIM / 2 + 3 * j + i] = sinf(sx + half_pi);
    }
}
static void posenc_slots_grad(const float *xyz, const float *g_out, float *g_xyz) {
  const float half_pi = (rn_f32)(0.5 * 3.14159265358979323846);
  for (int i = 0; i < 3; ++i) {
    float g = g_out[i];
    for (int j = 0; j < RN_POSENC_DEG; ++j) {
      float sc = (float)(1 << j), sx = xyz[i] * sc;
      g += sc * (cosf(sx) * g_out[3 + 3 * j + i] + cosf(sx + half_pi) * g_out[RN_IDE_DIM / 2 + 3 * j + i]);
    }
    g_xyz[i] = g;
  }
}

static int ide_term_index(int l, int m) { /* position of (l,m) in the 36-list */
  int base = 0;
  for (int li = 0; li < 5; ++li) { if (IDE_L[li] == l) return base + m; base += IDE_L[li] + 1; }
  return -1;
}
void rn_ide_stable_f32(const float *xyz, float kappa_inv, float *out) {
  if (!g_ide_init) { rn_ide_tables(g_ide_c, &g_ide_a[0][0], &g_ide_b[0][0]); g_ide_init = 1; }
  float x = xyz[0], y = xyz[1], z = xyz[2];
  float att[17];
  for (int li = 0; li < 5; ++li) {
    int l = IDE_L[li];
    att[l] = expf(-(float)(0.5 * l * (l + 1)) * kappa_inv);
  }
  float pr = 1.0f, pi = 0.0f;
  for (int m = 0; m <= 16; ++m) {
    if (m > 0) { float nr = pr * x - pi * y; float ni = pr * y + pi * x; pr = nr; pi = ni; }
    float tm2 = 0.0f, tm1 = g_ide_c[m]; /* T_{l-2}, T_{l-1} with l-1 = m */
    for (int l = m; l <= 16; ++l) {
      float tl;
      if (l == m) tl = g_ide_c[m];
      else { tl = g_ide_a[m][l] * (z * tm1 - g_ide_b[m][l] * tm2); tm2 = tm1; tm1 = tl; }
      if (l >= 1 && (l & (l - 1)) == 0) {
        int idx = ide_term_index(l, m);
        float s = tl * att[l];
        out[idx] = pr * s;
        out[RN_IDE_TERMS + idx] = pi * s;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* per-sample MLP                                                     */
/* ------------------------------------------------------------------ */

static float softplus_t(float x) { /* F.softplus beta=1 threshold=20 */
  return x > 20.0f ? x : log1pf(expf(x));
}
static float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }

/* image.py:51-59 */
float rn_linear_to_srgb(float x) {
  float srgb0 = (float)(323.0 / 25.0) * x;
  float srgb1 = (211.0f * powf(fmaxf(EPS32, x), (float)(5.0 / 12.0)) - 11.0f) / 200.0f;
  return (x <= 0.0031308f) ? srgb0 : srgb1;
}

/* ---- dense layers -------------------------------------------------------
 * y[s][o] = b[o] + sum_k W[o][k] x[s][k] as a k-ordered fmaf chain seeded with
 * the bias (nn.Linear = addmm(bias, x, W^T)).  Weights are held transposed
 * ([in][out]) so the chain vectorises across outputs; the per-output
 * summation order (k ascending, one rounding per fma) is unchanged. */
#define RN_SB 8 /* samples per block */

typedef struct rn_model {
  rn_param_offsets L;
  const float *P;    /* canonical blob */
  float *T;          /* transposed copies, same offsets, [in][out] */
} rn_model;

static void model_init(rn_model *m, const float *P) {
  rn_param_layout(&m->L);
  m->P = P;
  m->T = (float *)malloc(sizeof(float) * (size_t)m->L.total);
  memcpy(m->T, P, sizeof(float) * (size_t)m->L.total);
#define TR(off, out, in) do { for (int o_ = 0; o_ < (out); ++o_) for (int k_ = 0; k_ < (in); ++k_) m->T[(off) + (size_t)k_ * (out) + o_] = P[(off) + (size_t)o_ * (in) + k_]; } while (0)
  for (int i = 0; i < RN_DEPTH; ++i) { TR(m->L.sp_w[i], RN_WIDTH, m->L.sp_in[i]); TR(m->L.vd_w[i], RN_WIDTH, m->L.vd_in[i]); }
  TR(m->L.gradpred_w, 3, RN_WIDTH); TR(m->L.diffuse_w, 3, RN_WIDTH); TR(m->L.tint_w, 3, RN_WIDTH);
  TR(m->L.bneck_w, RN_BNECK, RN_WIDTH); TR(m->L.rgb_w, 3, RN_WIDTH);
#undef TR
}
static void model_free(rn_model *m) { free(m->T); }

/* X: [S][ldx], Y: [S][ldy]; Wt: [in][out]; out multiple of 16 uses AVX2. */
static void dense_block(const float *Wt, const float *b, const float *X, int ldx, int in,
                        int out, float *Y, int ldy, int S, int relu) {
#if defined(__AVX2__) && defined(__FMA__) && defined(RN_ORACLE_F64)
  if (out % 8 == 0) {                                  /* float64 build: the same k-ordered chains, 4 lanes per vector */
    for (int ob = 0; ob < out; ob += 8) {
      __m256d a0[RN_SB], a1[RN_SB];
      __m256d b0 = _mm256_loadu_pd(b + ob), b1 = _mm256_loadu_pd(b + ob + 4);
      for (int s = 0; s < S; ++s) { a0[s] = b0; a1[s] = b1; }
      for (int k = 0; k < in; ++k) {
        __m256d w0 = _mm256_loadu_pd(Wt + (size_t)k * out + ob);
        __m256d w1 = _mm256_loadu_pd(Wt + (size_t)k * out + ob + 4);
        for (int s = 0; s < S; ++s) {
          __m256d xs = _mm256_broadcast_sd(X + (size_t)s * ldx + k);
          a0[s] = _mm256_fmadd_pd(w0, xs, a0[s]);
          a1[s] = _mm256_fmadd_pd(w1, xs, a1[s]);
        }
      }
      __m256d z = _mm256_setzero_pd();
      for (int s = 0; s < S; ++s) {
        if (relu) { a0[s] = _mm256_max_pd(a0[s], z); a1[s] = _mm256_max_pd(a1[s], z); }
        _mm256_storeu_pd(Y + (size_t)s * ldy + ob, a0[s]);
        _mm256_storeu_pd(Y + (size_t)s * ldy + ob + 4, a1[s]);
      }
    }
    return;
  }
#elif defined(__AVX2__) && defined(__FMA__)
  if (out % 16 == 0) {
    for (int ob = 0; ob < out; ob += 16) {
      __m256 a0[RN_SB], a1[RN_SB];
      __m256 b0 = _mm256_loadu_ps(b + ob), b1 = _mm256_loadu_ps(b + ob + 8);
      for (int s = 0; s < S; ++s) { a0[s] = b0; a1[s] = b1; }
      for (int k = 0; k < in; ++k) {
        __m256 w0 = _mm256_loadu_ps(Wt + (size_t)k * out + ob);
        __m256 w1 = _mm256_loadu_ps(Wt + (size_t)k * out + ob + 8);
        for (int s = 0; s < S; ++s) {
          __m256 xs = _mm256_broadcast_ss(X + (size_t)s * ldx + k);
          a0[s] = _mm256_fmadd_ps(w0, xs, a0[s]);
          a1[s] = _mm256_fmadd_ps(w1, xs, a1[s]);
        }
      }
      __m256 z = _mm256_setzero_ps();
      for (int s = 0; s < S; ++s) {
        if (relu) { a0[s] = _mm256_max_ps(a0[s], z); a1[s] = _mm256_max_ps(a1[s], z); }
        _mm256_storeu_ps(Y + (size_t)s * ldy + ob, a0[s]);
        _mm256_storeu_ps(Y + (size_t)s * ldy + ob + 8, a1[s]);
      }
    }
    return;
  }
#endif
  for (int s = 0; s < S; ++s)
    for (int o = 0; o < out; ++o) {
      float a = b[o];
      for (int k = 0; k < in; ++k) a = fmaf(Wt[(size_t)k * out + o], X[(size_t)s * ldx + k], a);
      Y[(size_t)s * ldy + o] = relu ? fmaxf(a, 0.0f) : a;
    }
}

/* models.py:533-750 for a block of S <= RN_SB samples of one ray (Ref-NeRF flag
 * set of configs/blender_refnerf.gin:34-52). lmean/lvar: [S][3]. */
static void mlp_block(const rn_model *M, const rn_level_cfg *cfg, const float *lmean,
                      const float *lvar, const float *v, int S, rn_sample_out *outs) {
  const rn_param_offsets *L = &M->L;
  const float *P = M->P, *T = M->T;
  enum { LDX = RN_WIDTH + RN_DIR_IN };
  float ipe[RN_SB][RN_IPE_DIM];
  float x[RN_SB][LDX], y[RN_SB][RN_WIDTH];
  static __thread float act[RN_DEPTH][RN_SB][RN_WIDTH];     /* post-ReLU, for the VJP */
  for (int s = 0; s < S; ++s) {
    rn_ipe(lmean + 3 * s, lvar + 3 * s, ipe[s]);             /* models.py:566-571 */
    memcpy(x[s], ipe[s], sizeof(ipe[s]));
  }
  for (int i = 0; i < RN_DEPTH; ++i) {                       /* models.py:576-580 */
    dense_block(T + L->sp_w[i], P + L->sp_b[i], &x[0][0], LDX, L->sp_in[i], RN_WIDTH, &y[0][0], RN_WIDTH, S, 1);
    for (int s = 0; s < S; ++s) {
      memcpy(act[i][s], y[s], sizeof(y[s]));
      memcpy(x[s], y[s], sizeof(y[s]));
      if (i % RN_SKIP == 0 && i > 0) memcpy(x[s] + RN_WIDTH, ipe[s], sizeof(ipe[s]));
    }
  }
  float bneck[RN_SB][RN_BNECK];
  dense_block(T + L->bneck_w, P + L->bneck_b, &x[0][0], LDX, RN_WIDTH, RN_BNECK, &bneck[0][0], RN_BNECK, S, 0); /* :645 */
  for (int s = 0; s < S; ++s) {
    rn_sample_out *out = &outs[s];
    float raw_density, gp[3], raw_rough, raw_diff[3], raw_tint[3];
    dense_block(P + L->density_w, P + L->density_b, x[s], LDX, RN_WIDTH, 1, &raw_density, 1, 1, 0);  /* :582 */
    dense_block(T + L->gradpred_w, P + L->gradpred_b, x[s], LDX, RN_WIDTH, 3, gp, 3, 1, 0);          /* :613 */
    dense_block(T + L->diffuse_w, P + L->diffuse_b, x[s], LDX, RN_WIDTH, 3, raw_diff, 3, 1, 0);      /* :634 */
    dense_block(T + L->tint_w, P + L->tint_b, x[s], LDX, RN_WIDTH, 3, raw_tint, 3, 1, 0);            /* :637 */
    dense_block(P + L->rough_w, P + L->rough_b, x[s], LDX, RN_WIDTH, 1, &raw_rough, 1, 1, 0);        /* :640 */

    /* density-gradient normals (models.py:603-609), training only: VJP of
     * sum(raw_density) w.r.t. the (un-lifted) sample mean; detached. */
    out->normals[0] = out->normals[1] = out->normals[2] = 0.0f;
    if (cfg->training) {
      float g[RN_WIDTH + RN_IPE_DIM], gi[RN_WIDTH + RN_IPE_DIM], gipe[RN_IPE_DIM];
      memset(gipe, 0, sizeof(gipe));
      for (int k = 0; k < RN_WIDTH; ++k) g[k] = P[L->density_w + k];
      for (int i = RN_DEPTH - 1; i >= 0; --i) {
        int in = L->sp_in[i];
        const float *W = P + L->sp_w[i];
        for (int k = 0; k < in; ++k) gi[k] = 0.0f;
        for (int o = 0; o < RN_WIDTH; ++o) {
          if (!(act[i][s][o] > 0.0f)) continue;              /* relu'(0) = 0 */
          float go = g[o];
          const float *w = W + (size_t)o * in;
          for (int k = 0; k < in; ++k) gi[k] = fmaf(w[k], go, gi[k]);
        }
        if (i == 0) { for (int k = 0; k < RN_IPE_DIM; ++k) gipe[k] += gi[k]; }
        else {
          if (in > RN_WIDTH) for (int k = 0; k < RN_IPE_DIM; ++k) gipe[k] += gi[RN_WIDTH + k];
          memcpy(g, gi, sizeof(float) * RN_WIDTH);
        }
      }
      /* through the IPE (coord.py:119-126): d/dx sin(r(x)) = cos(r(x)) */
      const float HALF_PI = (rn_f32)(0.5 * M_PI);
      float gl[3] = {0, 0, 0};
      for (int j = 0; j < RN_IPE_DEG; ++j) {
        float sc = ldexpf(1.0f, j), sc2 = ldexpf(1.0f, 2 * j);
        for (int b = 0; b < 3; ++b) {
          float xx = lmean[3 * s + b] * sc;
          float e = expf(-0.5f * (lvar[3 * s + b] * sc2));
          gl[b] += ((gipe[j * 3 + b] * e) * cosf(safe_arg(xx))) * sc;
          gl[b] += ((gipe[48 + j * 3 + b] * e) * cosf(safe_arg(xx + HALF_PI))) * sc;
        }
      }
      float gx[3] = {-gl[2], -gl[1], -gl[0]};                 /* basis^T */
      float n2g = fmaxf((gx[0] * gx[0] + gx[1] * gx[1]) + gx[2] * gx[2], EPS32);
      float ng = sqrtf(n2g);
      for (int i = 0; i < 3; ++i) out->normals[i] = -(gx[i] / ng);
    }

    /* models.py:611-616 */
    float n2 = fmaxf((gp[0] * gp[0] + gp[1] * gp[1]) + gp[2] * gp[2], EPS32);
    float nrm = sqrtf(n2);
    float np_[3];
    for (int i = 0; i < 3; ++i) { np_[i] = -(gp[i] / nrm); out->normals_pred[i] = np_[i]; out->grad_pred[i] = gp[i]; }
    out->density = softplus_t(raw_density + cfg->density_bias);                  /* :623 */
    for (int i = 0; i < 3; ++i) out->tint[i] = sigmoid_t(raw_tint[i]);
    float rough = softplus_t(raw_rough + cfg->roughness_bias);                   /* :640-641 */
    out->roughness = rough;
    /* reflect(-viewdirs, normals_pred) (models.py:662-663, ref_utils.py:36-37) */
    float w[3] = {-v[0], -v[1], -v[2]};
    float dot = (np_[0] * w[0] + np_[1] * w[1]) + np_[2] * w[2];
    float refdir[3];
    for (int i = 0; i < 3; ++i) refdir[i] = (2.0f * dot) * np_[i] - w[i];
    float *din = x[s] + RN_WIDTH;                 /* [bottleneck | ide | n.v] (:686) */
    memcpy(din, bneck[s], sizeof(bneck[s]));
    if (cfg->ide_mode == 1) rn_ide_ref_f32(refdir, rough, din + RN_BNECK);
    else if (cfg->ide_mode == 2) rn_posenc_slots_f32(refdir, din + RN_BNECK);    /* :487-492: pos_enc, roughness unused */
    else rn_ide_stable_f32(refdir, rough, din + RN_BNECK);                        /* :665 */
    din[RN_DIR_IN - 1] = (np_[0] * v[0] + np_[1] * v[1]) + np_[2] * v[2];        /* :679-683 */
    /* raw diffuse is parked in `out` until the colour head below */
    out->diffuse[0] = raw_diff[0]; out->diffuse[1] = raw_diff[1]; out->diffuse[2] = raw_diff[2];
  }
  /* directional MLP (models.py:690-694); x[s][256..456] keeps the 201-vector */
  float xin[RN_SB][LDX];
  for (int s = 0; s < S; ++s) memcpy(xin[s], x[s] + RN_WIDTH, sizeof(float) * RN_DIR_IN);
  for (int i = 0; i < RN_DEPTH; ++i) {
    if (i == 0)
      dense_block(T + L->vd_w[i], P + L->vd_b[i], &xin[0][0], LDX, L->vd_in[i], RN_WIDTH, &y[0][0], RN_WIDTH, S, 1);
    else
      dense_block(T + L->vd_w[i], P + L->vd_b[i], &x[0][0], LDX, L->vd_in[i], RN_WIDTH, &y[0][0], RN_WIDTH, S, 1);
    for (int s = 0; s < S; ++s) memcpy(x[s], y[s], sizeof(y[s]));   /* skip input already sits at x[s]+256 */
  }
  const float LOG3 = 1.0986122886681098f;
  for (int s = 0; s < S; ++s) {
    rn_sample_out *out = &outs[s];
    float raw_rgb[3];
    dense_block(T + L->rgb_w, P + L->rgb_b, x[s], LDX, RN_WIDTH, 3, raw_rgb, 3, 1, 0);
    float spec_lin[3], diff_lin[3], rgb[3];
    for (int i = 0; i < 3; ++i) {
      float sg = sigmoid_t(cfg->rgb_premultiplier * raw_rgb[i] + cfg->rgb_bias);  /* :699-700 */
      diff_lin[i] = sigmoid_t(out->diffuse[i] - LOG3);                            /* :705-706 */
      spec_lin[i] = out->tint[i] * sg;                                            /* :708 */
      rgb[i] = spec_lin[i] + diff_lin[i];
    }
    if (cfg->srgb_mapping) {                                                      /* :712-723 */
      if (cfg->srgb_mapping_normalization) {
        float mxc = fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]);
        float norm = fmaxf(mxc, 1.0f);
        for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] / norm;
      }
      for (int i = 0; i < 3; ++i) {
        rgb[i] = clip01(rn_linear_to_srgb(rgb[i]));
        out->diffuse[i] = clip01(rn_linear_to_srgb(diff_lin[i]));
        out->specular[i] = clip01(rn_linear_to_srgb(spec_lin[i]));
      }
    } else {
      for (int i = 0; i < 3; ++i) { out->diffuse[i] = diff_lin[i]; out->specular[i] = spec_lin[i]; }
    }
    float pad_scale = (float)(1.0 + 2.0 * (double)cfg->rgb_padding);
    for (int i = 0; i < 3; ++i) out->rgb[i] = rgb[i] * pad_scale - cfg->rgb_padding; /* :729 */
  }
}

void rn_mlp_sample(const float *P, const rn_level_cfg *cfg, const float *lmean,
                   const float *lvar, const float *v, rn_sample_out *out) {
  rn_model M;
  model_init(&M, P);
  mlp_block(&M, cfg, lmean, lvar, v, 1, out);
  model_free(&M);
}

/* ------------------------------------------------------------------ */
/* compositing                                                        */
/* ------------------------------------------------------------------ */

/* render.py:132-149 */
void rn_alpha_weights(const float *density, const float *tdist, const float *dir,
                      int N, int opaque_background, float *weights) {
  float norm = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
  double cum = 0.0;
  for (int i = 0; i < N; ++i) {
    float t_delta = tdist[i + 1] - tdist[i];
    float delta = t_delta * norm;
    float dd = density[i] * delta;
    if (opaque_background && i == N - 1) dd = INFINITY;
    float alpha = 1.0f - expf(-dd);
    float trans = expf(-(float)cum);
    weights[i] = alpha * trans;
    cum += (double)dd;    /* torch.cumsum: double accumulator, fp32 outputs */
  }
}

static void render_map(int mode, float *rgb, float *dif, float *spc) {
  if (mode == RN_SRGB_NONE) return;
  if (mode == RN_SRGB_NORM_LINEAR || mode == RN_SRGB_NORM_SRGB) {
    float norm = fmaxf(fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]), 1.0f);
    for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] / norm;
  }
  for (int i = 0; i < 3; ++i) {
    if (mode == RN_SRGB_SRGB || mode == RN_SRGB_NORM_SRGB) {
      rgb[i] = rn_linear_to_srgb(rgb[i]); dif[i] = rn_linear_to_srgb(dif[i]); spc[i] = rn_linear_to_srgb(spc[i]);
    }
    rgb[i] = clip01(rgb[i]); dif[i] = clip01(dif[i]); spc[i] = clip01(spc[i]);
  }
}

/* stepfun.py:294-307 (weighted_percentile) + math.py:114-142 (interp, fp64) */
static void percentiles(const float *tdist, const float *w, int N, float bg_w, float far, double *out3) {
  int n = N + 2; /* knots */
  double *xp = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  double *fp = xp + n;
  xp[0] = 0.0;
  double acc = 0.0;
  for (int i = 0; i < N; ++i) { acc += (double)w[i]; xp[i + 1] = (double)fminf(1.0f, (float)acc); }
  (void)bg_w; /* weights_aug[-1] is dropped by integrate_weights */
  xp[N + 1] = 1.0;
  for (int i = 0; i <= N; ++i) fp[i] = (double)tdist[i];
  fp[N + 1] = (double)far;
  const float ps[3] = {5.0f / 100.0f, 50.0f / 100.0f, 95.0f / 100.0f};
  for (int q = 0; q < 3; ++q) {
    double x = (double)ps[q];
    int cnt = 0;
    for (int j = 0; j < n; ++j) cnt += (x >= xp[j]);
    int idx = cnt - 1; if (idx < 0) idx = 0; if (idx > n - 2) idx = n - 2;
    double m = (fp[idx + 1] - fp[idx]) / (xp[idx + 1] - xp[idx]);
    double b = fp[idx] - m * xp[idx];
    out3[q] = m * x + b;
  }
  free(xp);
}

/* render.py:152-254 (volumetric_rendering) for one ray: per-sample outputs `so`, weights `wts`, tdist `td`. */
static void composite_ray(const rn_level_cfg *cfg, const rn_sample_out *so, const float *wts, const float *td,
                          float far, int r, rn_level_out *out) {
  const int N = cfg->n_samples;
  float acc = 0.0f, rgb[3] = {0, 0, 0}, dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0};
  float dist = 0.0f, nrm[3] = {0, 0, 0}, nrp[3] = {0, 0, 0}, tnt[3] = {0, 0, 0}, rgh = 0.0f, logd = 0.0f;
  for (int i = 0; i < N; ++i) {
    float w = wts[i];
    acc += w;
    float tmid = 0.5f * (td[i] + td[i + 1]);
    dist += w * tmid;
    logd += w * logf(tmid);
    rgh += w * so[i].roughness;
    for (int c = 0; c < 3; ++c) {
      rgb[c] += w * so[i].rgb[c]; dif[c] += w * so[i].diffuse[c]; spc[c] += w * so[i].specular[c];
      nrm[c] += w * so[i].normals[c]; nrp[c] += w * so[i].normals_pred[c]; tnt[c] += w * so[i].tint[c];
    }
  }
  float bg_w = fmaxf(0.0f, 1.0f - acc);
  for (int c = 0; c < 3; ++c) { rgb[c] += bg_w * cfg->bg_rgb; dif[c] += bg_w * cfg->bg_rgb; spc[c] += bg_w * cfg->bg_rgb; }
  render_map(cfg->render_srgb_mode, rgb, dif, spc);
  if (out->r_rgb) memcpy(out->r_rgb + 3 * (size_t)r, rgb, 3 * sizeof(float));
  if (out->r_diffuse) memcpy(out->r_diffuse + 3 * (size_t)r, dif, 3 * sizeof(float));
  if (out->r_specular) memcpy(out->r_specular + 3 * (size_t)r, spc, 3 * sizeof(float));
  if (out->r_distance) out->r_distance[r] = dist;
  if (out->r_acc) out->r_acc[r] = acc;
  if (cfg->compute_extras) {
    if (out->r_normals && cfg->training) memcpy(out->r_normals + 3 * (size_t)r, nrm, 3 * sizeof(float));
    if (out->r_normals_pred) memcpy(out->r_normals_pred + 3 * (size_t)r, nrp, 3 * sizeof(float));
    if (out->r_tint) memcpy(out->r_tint + 3 * (size_t)r, tnt, 3 * sizeof(float));
    if (out->r_roughness) out->r_roughness[r] = rgh;
    if (out->r_distance_mean) {
      float e = expf(logd / fmaxf(EPS32, acc));
      if (isnan(e)) e = INFINITY;              /* nan_to_num(x, inf) */
      if (isinf(e)) e = e > 0 ? FLT_MAX : -FLT_MAX;
      out->r_distance_mean[r] = fminf(fmaxf(e, td[0]), td[N]);
    }
    if (out->r_percentiles) percentiles(td, wts, N, bg_w, far, out->r_percentiles + 3 * (size_t)r);
  }
}

/* Stage entry: compute_alpha_weights (render.py:132-149) + volumetric_rendering (render.py:152-254) on
 * caller-supplied per-sample values.  density / roughness [R,N]; tdist [R,N+1]; dirs [R,3]; far [R]; the
 * [R,N,3] tensors may be NULL (treated as zero).  cfg: n_samples, opaque_background, render_srgb_mode,
 * compute_extras, training (-> r_normals), bg_rgb.  Writes out->weights and the r_* fields. */
int rn_render_rays(const rn_level_cfg *cfg, int R, const float *density, const float *tdist, const float *dirs,
                   const float *far, const float *rgb, const float *diffuse, const float *specular,
                   const float *normals, const float *normals_pred, const float *roughness, const float *tint,
                   rn_level_out *out) {
  const int N = cfg->n_samples;
  if (N < 1) return -1;
  rn_sample_out *so = (rn_sample_out *)calloc((size_t)N, sizeof(rn_sample_out));
  float *wts = (float *)malloc(sizeof(float) * (size_t)N);
  for (int r = 0; r < R; ++r) {
    rn_alpha_weights(density + (size_t)r * N, tdist + (size_t)r * (N + 1), dirs + 3 * (size_t)r, N, cfg->opaque_background, wts);
    for (int i = 0; i < N; ++i) {
      size_t e = (size_t)r * N + i;
      so[i].roughness = roughness ? roughness[e] : 0.0f;
      for (int c = 0; c < 3; ++c) {
        so[i].rgb[c] = rgb ? rgb[3 * e + c] : 0.0f;
        so[i].diffuse[c] = diffuse ? diffuse[3 * e + c] : 0.0f;
        so[i].specular[c] = specular ? specular[3 * e + c] : 0.0f;
        so[i].normals[c] = normals ? normals[3 * e + c] : 0.0f;
        so[i].normals_pred[c] = normals_pred ? normals_pred[3 * e + c] : 0.0f;
        so[i].tint[c] = tint ? tint[3 * e + c] : 0.0f;
      }
    }
    if (out->weights) memcpy(out->weights + (size_t)r * N, wts, sizeof(float) * (size_t)N);
    composite_ray(cfg, so, wts, tdist + (size_t)r * (N + 1), far[r], r, out);
  }
  free(so); free(wts);
  return 0;
}

/* ------------------------------------------------------------------ */
/* one level                                                          */
/* ------------------------------------------------------------------ */

#define ST3(ptr, r, i, N, val3) do { if (ptr) { float *q_ = (ptr) + ((size_t)(r) * (N) + (i)) * 3; q_[0] = (val3)[0]; q_[1] = (val3)[1]; q_[2] = (val3)[2]; } } while (0)

int rn_level_forward(const float *params, const rn_level_cfg *cfg, const rn_rays *rays,
                     int R, const float *sdist_in, const float *weights_in,
                     rn_level_out *out, int n_threads) {
  const int N = cfg->n_samples, M = cfg->n_in;
  if (N <= 1) return -1;                           /* stepfun.py:234-235 */
  if (cfg->ray_shape != 0 && cfg->ray_shape != 1) return -2; /* render.py:126 */
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  rn_model model;
  model_init(&model, params);
#pragma omp parallel
  {
    float *logits = (float *)malloc(sizeof(float) * (size_t)(M + (N + 1) * 2 + N * 2));
    float *sd = logits + M, *td = sd + (N + 1), *dens = td + (N + 1), *wts = dens + N;
    rn_sample_out *so = (rn_sample_out *)malloc(sizeof(rn_sample_out) * (size_t)N);
    int32_t *bidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
#pragma omp for schedule(dynamic, 1)
    for (int r = 0; r < R; ++r) {
      const float *o = rays->origins + 3 * r, *d = rays->directions + 3 * r, *v = rays->viewdirs + 3 * r;
      float near = rays->near[r], far = rays->far[r], radius = rays->radii[r];
      /* models.py:200-215 */
      rn_resample_logits(sdist_in + (size_t)r * (M + 1), weights_in + (size_t)r * M, M, cfg->anneal, cfg->resample_padding, logits);
      rn_sample_intervals(sdist_in + (size_t)r * (M + 1), logits, M, N, cfg->s_near, cfg->s_far, sd, bidx);
      for (int i = 0; i <= N; ++i) td[i] = rn_s_to_t_fn(sd[i], near, far, cfg->raydist);          /* :218 */
      for (int i0 = 0; i0 < N; i0 += RN_SB) {                                     /* :221-241 */
        float lm[RN_SB][3], lv[RN_SB][3];
        int S = (N - i0 < RN_SB) ? (N - i0) : RN_SB;
        for (int s = 0; s < S; ++s) {
          rn_cast_sample(o, d, radius, td[i0 + s], td[i0 + s + 1], cfg->ray_shape, lm[s], lv[s], NULL);
          if (cfg->disable_integration) lv[s][0] = lv[s][1] = lv[s][2] = 0.0f;          /* models.py:228-231 */
        }
        mlp_block(&model, cfg, &lm[0][0], &lv[0][0], v, S, &so[i0]);
        for (int s = 0; s < S; ++s) dens[i0 + s] = so[i0 + s].density;
      }
      rn_alpha_weights(dens, td, d, N, cfg->opaque_background, wts);              /* :244-249 */
      /* ---- history ---- */
      if (out->sdist) memcpy(out->sdist + (size_t)r * (N + 1), sd, sizeof(float) * (N + 1));
      if (out->bin_idx) memcpy(out->bin_idx + (size_t)r * N, bidx, sizeof(int32_t) * N);
      for (int i = 0; i < N; ++i) {
        if (out->density) out->density[(size_t)r * N + i] = dens[i];
        if (out->weights) out->weights[(size_t)r * N + i] = wts[i];
        if (out->roughness) out->roughness[(size_t)r * N + i] = so[i].roughness;
        ST3(out->rgb, r, i, N, so[i].rgb);
        if (cfg->training) ST3(out->normals, r, i, N, so[i].normals);
        ST3(out->normals_pred, r, i, N, so[i].normals_pred);
        ST3(out->grad_pred, r, i, N, so[i].grad_pred);
        ST3(out->tint, r, i, N, so[i].tint);
        ST3(out->diffuse, r, i, N, so[i].diffuse);
        ST3(out->specular, r, i, N, so[i].specular);
      }
      /* ---- render.py:152-254 ---- */
      composite_ray(cfg, so, wts, td, far, r, out);
    }
    free(logits); free(so); free(bidx);
  }
  model_free(&model);
  return 0;
}

/* ================================================================== */
/* training: per-sample forward cache + backward       */
/* ================================================================== */

typedef struct rn_cache {
  float ipe[RN_IPE_DIM];
  float sp_act[RN_DEPTH][RN_WIDTH];           /* post-ReLU */
  float raw_density, raw_rough, gp[3], raw_diff[3], raw_tint[3];
  float din[RN_DIR_IN];
  float vd_act[RN_DEPTH][RN_WIDTH];
  float raw_rgb[3];
  float refdir[3], rough, npred[3], nrm2, tint[3];
} rn_cache;

static float softplus_grad(float x) { return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)); }

/* gradient of rn_ide_stable_f32 w.r.t. xyz and kappa_inv given g_out[72] */
static void ide_stable_grad(const float *xyz, float kappa_inv, const float *g_out, float *g_xyz, float *g_kappa) {
  if (!g_ide_init) { rn_ide_tables(g_ide_c, &g_ide_a[0][0], &g_ide_b[0][0]); g_ide_init = 1; }
  float x = xyz[0], y = xyz[1], z = xyz[2];
  float pr[17], pi[17], gpr[17], gpi[17];
  pr[0] = 1.0f; pi[0] = 0.0f;
  for (int m = 1; m <= 16; ++m) { pr[m] = pr[m - 1] * x - pi[m - 1] * y; pi[m] = pr[m - 1] * y + pi[m - 1] * x; }
  for (int m = 0; m <= 16; ++m) gpr[m] = gpi[m] = 0.0f;
  float gz = 0.0f, gk = 0.0f;
  for (int m = 0; m <= 16; ++m) {
    float tm2 = 0.0f, tm1 = g_ide_c[m], dm2 = 0.0f, dm1 = 0.0f;
    for (int l = m; l <= 16; ++l) {
      float tl, dl;
      if (l == m) { tl = g_ide_c[m]; dl = 0.0f; }
      else {
        float a = g_ide_a[m][l], b = g_ide_b[m][l];
        tl = a * (z * tm1 - b * tm2);
        dl = a * (tm1 + z * dm1 - b * dm2);
        tm2 = tm1; tm1 = tl; dm2 = dm1; dm1 = dl;
      }
      if (l >= 1 && (l & (l - 1)) == 0) {
        int idx = ide_term_index(l, m);
        float sigma = (float)(0.5 * l * (l + 1));
        float att = expf(-sigma * kappa_inv);
        float gre = g_out[idx], gim = g_out[RN_IDE_TERMS + idx];
        float s = tl * att;
        float gs = gre * pr[m] + gim * pi[m];
        gpr[m] += gre * s; gpi[m] += gim * s;
        gz += gs * att * dl;
        gk += gs * tl * (-sigma * att);
      }
    }
  }
  float gx = 0.0f, gy = 0.0f;
  for (int m = 1; m <= 16; ++m) {
    gx += (float)m * (gpr[m] * pr[m - 1] + gpi[m] * pi[m - 1]);
    gy += (float)m * (-gpr[m] * pi[m - 1] + gpi[m] * pr[m - 1]);
  }
  g_xyz[0] = gx; g_xyz[1] = gy; g_xyz[2] = gz;
  *g_kappa = gk;
}

static float srgb_grad(float u) { /* d linear_to_srgb / du (image.py:51-59) */
  if (u <= 0.0031308f) return (float)(323.0 / 25.0);
  if (!(u > EPS32)) return 0.0f;
  return (211.0f / 200.0f) * (float)(5.0 / 12.0) * powf(u, (float)(5.0 / 12.0) - 1.0f);
}

/* forward of one sample with everything the backward needs */
static void mlp_forward_cached(const rn_model *M, const rn_level_cfg *cfg, const float *lmean, const float *lvar,
                               const float *v, rn_cache *c, rn_sample_out *out) {
  const rn_param_offsets *L = &M->L;
  const float *P = M->P, *T = M->T;
  float x[RN_WIDTH + RN_DIR_IN], y[RN_WIDTH];
  rn_ipe(lmean, lvar, c->ipe);
  memcpy(x, c->ipe, sizeof(c->ipe));
  for (int i = 0; i < RN_DEPTH; ++i) {
    dense_block(T + L->sp_w[i], P + L->sp_b[i], x, 0, L->sp_in[i], RN_WIDTH, y, RN_WIDTH, 1, 1);
    memcpy(c->sp_act[i], y, sizeof(y));
    memcpy(x, y, sizeof(y));
    if (i % RN_SKIP == 0 && i > 0) memcpy(x + RN_WIDTH, c->ipe, sizeof(c->ipe));
  }
  float bneck[RN_BNECK];
  dense_block(T + L->bneck_w, P + L->bneck_b, x, 0, RN_WIDTH, RN_BNECK, bneck, RN_BNECK, 1, 0);
  dense_block(P + L->density_w, P + L->density_b, x, 0, RN_WIDTH, 1, &c->raw_density, 1, 1, 0);
  dense_block(T + L->gradpred_w, P + L->gradpred_b, x, 0, RN_WIDTH, 3, c->gp, 3, 1, 0);
  dense_block(T + L->diffuse_w, P + L->diffuse_b, x, 0, RN_WIDTH, 3, c->raw_diff, 3, 1, 0);
  dense_block(T + L->tint_w, P + L->tint_b, x, 0, RN_WIDTH, 3, c->raw_tint, 3, 1, 0);
  dense_block(P + L->rough_w, P + L->rough_b, x, 0, RN_WIDTH, 1, &c->raw_rough, 1, 1, 0);
  c->nrm2 = (c->gp[0] * c->gp[0] + c->gp[1] * c->gp[1]) + c->gp[2] * c->gp[2];
  float nrm = sqrtf(fmaxf(c->nrm2, EPS32));
  for (int i = 0; i < 3; ++i) { c->npred[i] = -(c->gp[i] / nrm); c->tint[i] = sigmoid_t(c->raw_tint[i]); }
  c->rough = softplus_t(c->raw_rough + cfg->roughness_bias);
  float w[3] = {-v[0], -v[1], -v[2]};
  float dot = (c->npred[0] * w[0] + c->npred[1] * w[1]) + c->npred[2] * w[2];
  for (int i = 0; i < 3; ++i) c->refdir[i] = (2.0f * dot) * c->npred[i] - w[i];
  memcpy(c->din, bneck, sizeof(bneck));
  if (cfg->ide_mode == 2) rn_posenc_slots_f32(c->refdir, c->din + RN_BNECK);
  else rn_ide_stable_f32(c->refdir, c->rough, c->din + RN_BNECK);
  c->din[RN_DIR_IN - 1] = (c->npred[0] * v[0] + c->npred[1] * v[1]) + c->npred[2] * v[2];
  memcpy(x, c->din, sizeof(c->din));
  for (int i = 0; i < RN_DEPTH; ++i) {
    dense_block(T + L->vd_w[i], P + L->vd_b[i], x, 0, L->vd_in[i], RN_WIDTH, y, RN_WIDTH, 1, 1);
    memcpy(c->vd_act[i], y, sizeof(y));
    memcpy(x, y, sizeof(y));
    if (i % RN_SKIP == 0 && i > 0) memcpy(x + RN_WIDTH, c->din, sizeof(c->din));
  }
  dense_block(T + L->rgb_w, P + L->rgb_b, x, 0, RN_WIDTH, 3, c->raw_rgb, 3, 1, 0);
  (void)out;
}

/* acc[o][k] += d[o] * x[k], bias grad, and gin[k] = sum_o W[o][k] d[o] */
static void dense_backward(const float *W, int in, int out, const float *d, const float *xin, float *gW, float *gb, float *gin) {
  if (gin) for (int k = 0; k < in; ++k) gin[k] = 0.0f;
  for (int o = 0; o < out; ++o) {
    float dv = d[o];
    if (dv == 0.0f) continue;
    const float *w = W + (size_t)o * in;
    float *gw = gW + (size_t)o * in;
    for (int k = 0; k < in; ++k) gw[k] += dv * xin[k];
    gb[o] += dv;
    if (gin) for (int k = 0; k < in; ++k) gin[k] += w[k] * dv;
  }
}

/* backward of one sample: upstream grads w.r.t. density, sample rgb, n_pred */
/* torch.clip(linear_to_srgb(x), 0, 1) backward for one channel (models.py:718-719, render.py:204-206) */
static float srgb_clip_grad(float x) {
  float yv = rn_linear_to_srgb(x);
  return (yv >= 0.0f && yv <= 1.0f) ? srgb_grad(x) : 0.0f;
}

/* extra[0..2] = dL/d history diffuse, [3..5] = dL/d history specular, [6..8] = dL/d history tint,
 * [9] = dL/d history roughness (NULL = none). */
static void mlp_backward_x(const rn_model *M, const rn_level_cfg *cfg, const rn_cache *c, const float *v,
                           float g_density, const float *g_rgb_out, const float *g_npred_loss,
                           const float *extra, float *G);
static void mlp_backward(const rn_model *M, const rn_level_cfg *cfg, const rn_cache *c, const float *v,
                         float g_density, const float *g_rgb_out, const float *g_npred_loss, float *G) {
  mlp_backward_x(M, cfg, c, v, g_density, g_rgb_out, g_npred_loss, NULL, G);
}
static void mlp_backward_x(const rn_model *M, const rn_level_cfg *cfg, const rn_cache *c, const float *v,
                           float g_density, const float *g_rgb_out, const float *g_npred_loss,
                           const float *extra, float *G) {
  const rn_param_offsets *L = &M->L;
  const float *P = M->P;
  const float LOG3 = 1.0986122886681098f;
  /* ---- colour head (models.py:699-729) ---- */
  float sg[3], dl[3], spec[3], col[3];
  for (int i = 0; i < 3; ++i) {
    sg[i] = sigmoid_t(cfg->rgb_premultiplier * c->raw_rgb[i] + cfg->rgb_bias);
    dl[i] = sigmoid_t(c->raw_diff[i] - LOG3);
    spec[i] = c->tint[i] * sg[i];
    col[i] = spec[i] + dl[i];
  }
  float pad_scale = (float)(1.0 + 2.0 * (double)cfg->rgb_padding);
  float g_col[3];
  for (int i = 0; i < 3; ++i) g_col[i] = g_rgb_out[i] * pad_scale;
  if (cfg->srgb_mapping) {
    float u[3], norm = 1.0f, mxc = fmaxf(fmaxf(col[0], col[1]), col[2]);
    if (cfg->srgb_mapping_normalization) norm = fmaxf(mxc, 1.0f);
    float g_u[3];
    for (int i = 0; i < 3; ++i) {
      u[i] = col[i] / norm;
      float yv = rn_linear_to_srgb(u[i]);
      float pass = (yv >= 0.0f && yv <= 1.0f) ? 1.0f : 0.0f;   /* torch.clip backward */
      g_u[i] = g_col[i] * pass * srgb_grad(u[i]);
    }
    float g_norm = 0.0f;
    for (int i = 0; i < 3; ++i) { g_col[i] = g_u[i] / norm; g_norm += -g_u[i] * col[i] / (norm * norm); }
    if (cfg->srgb_mapping_normalization) {
      float gm = (mxc > 1.0f) ? g_norm : (mxc == 1.0f ? 0.5f * g_norm : 0.0f);   /* torch.maximum */
      if (gm != 0.0f) {
        int cnt = 0;
        for (int i = 0; i < 3; ++i) cnt += (col[i] == mxc);
        for (int i = 0; i < 3; ++i) if (col[i] == mxc) g_col[i] += gm / (float)cnt;     /* amax: ties share */
      }
    }
  }
  float g_tint[3], g_raw_rgb[3], g_raw_diff[3];
  for (int i = 0; i < 3; ++i) {
    float g_dl = g_col[i], g_sp = g_col[i];
    if (extra) {   /* the history's own diffuse / specular: clip(srgb(.)) of the linear colours, or the colours */
      g_dl += extra[i] * (cfg->srgb_mapping ? srgb_clip_grad(dl[i]) : 1.0f);
      g_sp += extra[3 + i] * (cfg->srgb_mapping ? srgb_clip_grad(spec[i]) : 1.0f);
    }
    g_tint[i] = g_sp * sg[i] + (extra ? extra[6 + i] : 0.0f);
    g_raw_rgb[i] = (g_sp * c->tint[i]) * sg[i] * (1.0f - sg[i]) * cfg->rgb_premultiplier;
    g_raw_diff[i] = g_dl * dl[i] * (1.0f - dl[i]);
  }
  /* ---- directional MLP ---- */
  float d[RN_WIDTH + RN_DIR_IN], gin[RN_WIDTH + RN_DIR_IN], g_din[RN_DIR_IN];
  float xin[RN_WIDTH + RN_DIR_IN];
  memset(g_din, 0, sizeof(g_din));
  dense_backward(P + L->rgb_w, RN_WIDTH, 3, g_raw_rgb, c->vd_act[RN_DEPTH - 1], G + L->rgb_w, G + L->rgb_b, gin);
  for (int i = RN_DEPTH - 1; i >= 0; --i) {
    for (int o = 0; o < RN_WIDTH; ++o) d[o] = (c->vd_act[i][o] > 0.0f) ? gin[o] : 0.0f;
    int in = L->vd_in[i];
    if (i == 0) memcpy(xin, c->din, sizeof(c->din));
    else { memcpy(xin, c->vd_act[i - 1], sizeof(float) * RN_WIDTH); if (in > RN_WIDTH) memcpy(xin + RN_WIDTH, c->din, sizeof(c->din)); }
    dense_backward(P + L->vd_w[i], in, RN_WIDTH, d, xin, G + L->vd_w[i], G + L->vd_b[i], gin);
    if (i == 0) { for (int k = 0; k < RN_DIR_IN; ++k) g_din[k] += gin[k]; }
    else if (in > RN_WIDTH) { for (int k = 0; k < RN_DIR_IN; ++k) g_din[k] += gin[RN_WIDTH + k]; }
  }
  /* ---- encodings / reflection (models.py:657-686) ---- */
  float g_ref[3], g_rough, g_np[3];
  if (cfg->ide_mode == 2) { posenc_slots_grad(c->refdir, g_din + RN_BNECK, g_ref); g_rough = 0.0f; }
  else ide_stable_grad(c->refdir, c->rough, g_din + RN_BNECK, g_ref, &g_rough);
  float g_dot = g_din[RN_DIR_IN - 1];
  float w[3] = {-v[0], -v[1], -v[2]};
  float ndw = (c->npred[0] * w[0] + c->npred[1] * w[1]) + c->npred[2] * w[2];
  float grn = (g_ref[0] * c->npred[0] + g_ref[1] * c->npred[1]) + g_ref[2] * c->npred[2];
  for (int i = 0; i < 3; ++i) g_np[i] = g_npred_loss[i] + 2.0f * (grn * w[i] + ndw * g_ref[i]) + g_dot * v[i];
  /* n_pred = -g/sqrt(max(|g|^2, eps)) (ref_utils.py:40-42) */
  float s = fmaxf(c->nrm2, EPS32), rs = sqrtf(s);
  float g_gp[3];
  {
    float gdotg = (c->gp[0] * g_np[0] + c->gp[1] * g_np[1]) + c->gp[2] * g_np[2];
    float live = (c->nrm2 > EPS32) ? 1.0f : (c->nrm2 == EPS32 ? 0.5f : 0.0f);
    for (int i = 0; i < 3; ++i) g_gp[i] = -(g_np[i] / rs - live * c->gp[i] * gdotg / (s * rs));
  }
  if (extra) g_rough += extra[9];
  float g_raw_rough = g_rough * softplus_grad(c->raw_rough + cfg->roughness_bias);
  float g_raw_tint[3];
  for (int i = 0; i < 3; ++i) g_raw_tint[i] = g_tint[i] * c->tint[i] * (1.0f - c->tint[i]);
  float g_raw_density = g_density * softplus_grad(c->raw_density + cfg->density_bias);
  /* ---- heads ---- */
  const float *x7 = c->sp_act[RN_DEPTH - 1];
  float gx[RN_WIDTH], tmp[RN_WIDTH + RN_IPE_DIM];
  memset(gx, 0, sizeof(gx));
#define HEADB(woff, boff, outn, dvec) do { dense_backward(P + (woff), RN_WIDTH, (outn), (dvec), x7, G + (woff), G + (boff), tmp); for (int k_ = 0; k_ < RN_WIDTH; ++k_) gx[k_] += tmp[k_]; } while (0)
  HEADB(L->density_w, L->density_b, 1, &g_raw_density);
  HEADB(L->gradpred_w, L->gradpred_b, 3, g_gp);
  HEADB(L->rough_w, L->rough_b, 1, &g_raw_rough);
  HEADB(L->diffuse_w, L->diffuse_b, 3, g_raw_diff);
  HEADB(L->tint_w, L->tint_b, 3, g_raw_tint);
  HEADB(L->bneck_w, L->bneck_b, RN_BNECK, g_din);
#undef HEADB
  /* ---- spatial MLP ---- */
  memcpy(gin, gx, sizeof(gx));
  for (int i = RN_DEPTH - 1; i >= 0; --i) {
    for (int o = 0; o < RN_WIDTH; ++o) d[o] = (c->sp_act[i][o] > 0.0f) ? gin[o] : 0.0f;
    int in = L->sp_in[i];
    if (i == 0) memcpy(xin, c->ipe, sizeof(c->ipe));
    else { memcpy(xin, c->sp_act[i - 1], sizeof(float) * RN_WIDTH); if (in > RN_WIDTH) memcpy(xin + RN_WIDTH, c->ipe, sizeof(c->ipe)); }
    dense_backward(P + L->sp_w[i], in, RN_WIDTH, d, xin, G + L->sp_w[i], G + L->sp_b[i], (i > 0) ? gin : NULL);
  }
}

int rn_level_train(const float *params, const rn_level_cfg *cfg_in, const rn_rays *rays, int R,
                   const float *sdist_in, const float *weights_in, const float *gt_rgb,
                   const float *lossmult, const rn_loss_cfg *lc, rn_level_out *out,
                   float *grads, double *loss3, int n_threads) {
  rn_level_cfg cfgv = *cfg_in;
  cfgv.training = 1;
  const rn_level_cfg *cfg = &cfgv;
  const int N = cfg->n_samples, M = cfg->n_in;
  if (N <= 1) return -1;
  rn_model model;
  model_init(&model, params);
  double denom = 0.0;
  for (int r = 0; r < R; ++r) denom += 3.0 * (double)lossmult[r];
  double l_data = 0.0, l_or = 0.0, l_nm = 0.0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel reduction(+ : l_data, l_or, l_nm)
  {
    float *G = (float *)calloc((size_t)model.L.total, sizeof(float));
    rn_cache *cache = (rn_cache *)malloc(sizeof(rn_cache) * (size_t)N);
    rn_sample_out *so = (rn_sample_out *)malloc(sizeof(rn_sample_out) * (size_t)N);
    float *buf = (float *)malloc(sizeof(float) * (size_t)(M + 5 * (N + 1)));
    float *logits = buf, *sd = buf + M, *td = sd + (N + 1), *dens = td + (N + 1), *wts = dens + (N + 1), *gw = wts + (N + 1);
    int32_t *bidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
#pragma omp for schedule(dynamic, 1)
    for (int r = 0; r < R; ++r) {
      const float *o = rays->origins + 3 * r, *d = rays->directions + 3 * r, *v = rays->viewdirs + 3 * r;
      float nearv = rays->near[r], farv = rays->far[r], radius = rays->radii[r];
      rn_resample_logits(sdist_in + (size_t)r * (M + 1), weights_in + (size_t)r * M, M, cfg->anneal, cfg->resample_padding, logits);
      rn_sample_intervals(sdist_in + (size_t)r * (M + 1), logits, M, N, cfg->s_near, cfg->s_far, sd, bidx);
      for (int i = 0; i <= N; ++i) td[i] = rn_s_to_t_fn(sd[i], nearv, farv, cfg->raydist);
      for (int i = 0; i < N; ++i) {
        float lm[3], lv[3];
        rn_cast_sample(o, d, radius, td[i], td[i + 1], cfg->ray_shape, lm, lv, NULL);
        if (cfg->disable_integration) lv[0] = lv[1] = lv[2] = 0.0f;
        mlp_block(&model, cfg, lm, lv, v, 1, &so[i]);              /* outputs incl. density normals */
        mlp_forward_cached(&model, cfg, lm, lv, v, &cache[i], NULL);
        dens[i] = so[i].density;
      }
      rn_alpha_weights(dens, td, d, N, cfg->opaque_background, wts);
      if (out) {
        if (out->sdist) memcpy(out->sdist + (size_t)r * (N + 1), sd, sizeof(float) * (N + 1));
        if (out->weights) memcpy(out->weights + (size_t)r * N, wts, sizeof(float) * N);
      }
      /* ---- forward render + losses ---- */
      float acc = 0.0f, rgb[3] = {0, 0, 0};
      for (int i = 0; i < N; ++i) { acc += wts[i]; for (int c = 0; c < 3; ++c) rgb[c] += wts[i] * so[i].rgb[c]; }
      float bg_w = fmaxf(0.0f, 1.0f - acc);
      float pre[3];
      for (int c = 0; c < 3; ++c) { rgb[c] += bg_w * cfg->bg_rgb; pre[c] = rgb[c]; }
      float dif_[3] = {0, 0, 0}, spc_[3] = {0, 0, 0};
      render_map(cfg->render_srgb_mode, rgb, dif_, spc_);
      if (out && out->r_rgb) memcpy(out->r_rgb + 3 * (size_t)r, rgb, 3 * sizeof(float));
      float g_rgb[3];
      for (int c = 0; c < 3; ++c) {
        float res = rgb[c] - gt_rgb[3 * r + c];
        l_data += (double)lc->data_mult * (double)lossmult[r] * (double)res * (double)res / denom;
        g_rgb[c] = (float)((double)lc->data_mult * 2.0 * (double)lossmult[r] * (double)res / denom);
      }
      /* back through the render-time mapping (render.py:186-216) */
      {
        int mode = cfg->render_srgb_mode;
        if (mode != RN_SRGB_NONE) {
          float norm = 1.0f, mxc = fmaxf(fmaxf(pre[0], pre[1]), pre[2]);
          int normed = (mode == RN_SRGB_NORM_LINEAR || mode == RN_SRGB_NORM_SRGB);
          if (normed) norm = fmaxf(mxc, 1.0f);
          float gu[3], gnorm = 0.0f;
          for (int c = 0; c < 3; ++c) {
            float u = pre[c] / norm;
            float yv = (mode == RN_SRGB_SRGB || mode == RN_SRGB_NORM_SRGB) ? rn_linear_to_srgb(u) : u;
            float pass = (yv >= 0.0f && yv <= 1.0f) ? 1.0f : 0.0f;
            float dy = (mode == RN_SRGB_SRGB || mode == RN_SRGB_NORM_SRGB) ? srgb_grad(u) : 1.0f;
            gu[c] = g_rgb[c] * pass * dy;
          }
          for (int c = 0; c < 3; ++c) { g_rgb[c] = gu[c] / norm; gnorm += -gu[c] * pre[c] / (norm * norm); }
          if (normed) {
            float gm = (mxc > 1.0f) ? gnorm : (mxc == 1.0f ? 0.5f * gnorm : 0.0f);
            if (gm != 0.0f) {
              int cnt = 0;
              for (int c = 0; c < 3; ++c) cnt += (pre[c] == mxc);
              for (int c = 0; c < 3; ++c) if (pre[c] == mxc) g_rgb[c] += gm / (float)cnt;
            }
          }
        }
      }
      /* per-sample upstream gradients */
      float gsum = (g_rgb[0] + g_rgb[1] + g_rgb[2]) * cfg->bg_rgb;
      for (int i = 0; i < N; ++i) {
        float ndv = -((so[i].normals_pred[0] * v[0] + so[i].normals_pred[1] * v[1]) + so[i].normals_pred[2] * v[2]);
        float mn = fminf(0.0f, ndv);
        float nn = (so[i].normals[0] * so[i].normals_pred[0] + so[i].normals[1] * so[i].normals_pred[1]) + so[i].normals[2] * so[i].normals_pred[2];
        l_or += (double)lc->orientation_mult * (double)(wts[i] * mn * mn) / (double)R;
        l_nm += (double)lc->normal_mult * (double)(wts[i] * (1.0f - nn)) / (double)R;
        float g = (g_rgb[0] * so[i].rgb[0] + g_rgb[1] * so[i].rgb[1]) + g_rgb[2] * so[i].rgb[2];
        if (acc < 1.0f) g -= gsum;                      /* bg_w = max(0, 1 - acc) */
        g += lc->orientation_mult * (mn * mn) / (float)R + lc->normal_mult * (1.0f - nn) / (float)R;
        gw[i] = g;
      }
      /* weights -> density (render.py:132-149): w_i = (1-e^{-dd_i}) e^{-c_i} */
      float norm_d = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
      double suffix = 0.0;                               /* sum_{j>i} g_w_j w_j */
      {
        double cum_all = 0.0;
        for (int i = 0; i < N; ++i) cum_all += (double)(dens[i] * ((td[i + 1] - td[i]) * norm_d));
        double cum = cum_all;
        for (int i = N - 1; i >= 0; --i) {
          float delta = (td[i + 1] - td[i]) * norm_d;
          float dd = dens[i] * delta;
          cum -= (double)dd;                             /* c_i */
          float e_dd = expf(-dd), tr = expf(-(float)cum);
          float g_dd = gw[i] * e_dd * tr - (float)suffix;
          suffix += (double)(gw[i] * wts[i]);
          float g_density = g_dd * delta;
          float g_rgb_s[3], g_np[3];
          float ndv = -((so[i].normals_pred[0] * v[0] + so[i].normals_pred[1] * v[1]) + so[i].normals_pred[2] * v[2]);
          float mn = fminf(0.0f, ndv);
          for (int c = 0; c < 3; ++c) {
            g_rgb_s[c] = wts[i] * g_rgb[c];
            g_np[c] = lc->orientation_mult / (float)R * wts[i] * 2.0f * mn * (-v[c])
                      + lc->normal_mult / (float)R * wts[i] * (-so[i].normals[c]);
          }
          if (cfg->opaque_background && i == N - 1) g_density = 0.0f;
          mlp_backward(&model, cfg, &cache[i], v, g_density, g_rgb_s, g_np, G);
        }
      }
    }
#pragma omp critical
    { for (int k = 0; k < model.L.total; ++k) grads[k] += G[k]; }
    free(G); free(cache); free(so); free(buf); free(bidx);
  }
  model_free(&model);
  if (loss3) { loss3[0] = l_data; loss3[1] = l_or; loss3[2] = l_nm; }
  return 0;
}

/* ---- generic backward of one level: arbitrary upstream gradients on every differentiable output ----
 * (what autograd does for any loss written on renderings / ray_history: train_utils.py:207-329) */
static void render_map_backward(int mode, int allow_norm, const float *pre, float *g) {
  if (mode == RN_SRGB_NONE) return;
  int srgb = (mode == RN_SRGB_SRGB || mode == RN_SRGB_NORM_SRGB);
  int normed = allow_norm && (mode == RN_SRGB_NORM_LINEAR || mode == RN_SRGB_NORM_SRGB);
  float norm = 1.0f, mxc = fmaxf(fmaxf(pre[0], pre[1]), pre[2]);
  if (normed) norm = fmaxf(mxc, 1.0f);
  float gu[3], gnorm = 0.0f;
  for (int c = 0; c < 3; ++c) {
    float u = pre[c] / norm;
    float yv = srgb ? rn_linear_to_srgb(u) : u;
    float pass = (yv >= 0.0f && yv <= 1.0f) ? 1.0f : 0.0f;
    gu[c] = g[c] * pass * (srgb ? srgb_grad(u) : 1.0f);
  }
  for (int c = 0; c < 3; ++c) { g[c] = gu[c] / norm; gnorm += -gu[c] * pre[c] / (norm * norm); }
  if (normed) {
    float gm = (mxc > 1.0f) ? gnorm : (mxc == 1.0f ? 0.5f * gnorm : 0.0f);
    if (gm != 0.0f) {
      int cnt = 0;
      for (int c = 0; c < 3; ++c) cnt += (pre[c] == mxc);
      for (int c = 0; c < 3; ++c) if (pre[c] == mxc) g[c] += gm / (float)cnt;
    }
  }
}

static float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

int rn_level_backward(const float *params, const rn_level_cfg *cfg_in, const rn_rays *rays, int R,
                      const float *sdist_in, const float *weights_in, const rn_level_seeds *sd_,
                      float *grads, int n_threads) {
  rn_level_cfg cfgv = *cfg_in;
  cfgv.training = 1;
  const rn_level_cfg *cfg = &cfgv;
  const int N = cfg->n_samples, M = cfg->n_in;
  if (N <= 1) return -1;
  rn_model model;
  model_init(&model, params);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
  {
    float *G = (float *)calloc((size_t)model.L.total, sizeof(float));
    rn_cache *cache = (rn_cache *)malloc(sizeof(rn_cache) * (size_t)N);
    rn_sample_out *so = (rn_sample_out *)malloc(sizeof(rn_sample_out) * (size_t)N);
    float *buf = (float *)malloc(sizeof(float) * (size_t)(M + 5 * (N + 1)));
    float *logits = buf, *sd = buf + M, *td = sd + (N + 1), *dens = td + (N + 1), *wts = dens + (N + 1), *gw = wts + (N + 1);
#pragma omp for schedule(dynamic, 1)
    for (int r = 0; r < R; ++r) {
      const float *o = rays->origins + 3 * r, *d = rays->directions + 3 * r, *v = rays->viewdirs + 3 * r;
      float nearv = rays->near[r], farv = rays->far[r], radius = rays->radii[r];
      rn_resample_logits(sdist_in + (size_t)r * (M + 1), weights_in + (size_t)r * M, M, cfg->anneal, cfg->resample_padding, logits);
      rn_sample_intervals(sdist_in + (size_t)r * (M + 1), logits, M, N, cfg->s_near, cfg->s_far, sd, NULL);
      for (int i = 0; i <= N; ++i) td[i] = rn_s_to_t_fn(sd[i], nearv, farv, cfg->raydist);
      for (int i = 0; i < N; ++i) {
        float lm[3], lv[3];
        rn_cast_sample(o, d, radius, td[i], td[i + 1], cfg->ray_shape, lm, lv, NULL);
        if (cfg->disable_integration) lv[0] = lv[1] = lv[2] = 0.0f;
        mlp_block(&model, cfg, lm, lv, v, 1, &so[i]);
        mlp_forward_cached(&model, cfg, lm, lv, v, &cache[i], NULL);
        dens[i] = so[i].density;
      }
      rn_alpha_weights(dens, td, d, N, cfg->opaque_background, wts);
      /* composites before the render-time map (render.py:161-165) */
      float acc = 0.0f, pre[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 0; i < N; ++i) {
        acc += wts[i];
        for (int c = 0; c < 3; ++c) {
          pre[0][c] += wts[i] * so[i].rgb[c];
          pre[1][c] += wts[i] * so[i].diffuse[c];
          pre[2][c] += wts[i] * so[i].specular[c];
        }
      }
      float bg_w = fmaxf(0.0f, 1.0f - acc);
      float g3[3][3];
      const float *src3[3] = {sd_->g_r_rgb, sd_->g_r_diffuse, sd_->g_r_specular};
      float gsum = 0.0f;
      for (int k = 0; k < 3; ++k) {
        for (int c = 0; c < 3; ++c) { pre[k][c] += bg_w * cfg->bg_rgb; g3[k][c] = src3[k] ? src3[k][3 * r + c] : 0.0f; }
        render_map_backward(cfg->render_srgb_mode, k == 0, pre[k], g3[k]);
        gsum += ((g3[k][0] + g3[k][1]) + g3[k][2]) * cfg->bg_rgb;
      }
      const float g_acc = sd_->g_r_acc ? sd_->g_r_acc[r] : 0.0f, g_dist = sd_->g_r_distance ? sd_->g_r_distance[r] : 0.0f;
      const float zero3[3] = {0, 0, 0};
      const float *g_rn = sd_->g_r_normals ? sd_->g_r_normals + 3 * r : zero3;
      const float *g_rnp = sd_->g_r_normals_pred ? sd_->g_r_normals_pred + 3 * r : zero3;
      const float *g_rt = sd_->g_r_tint ? sd_->g_r_tint + 3 * r : zero3;
      const float g_rr = sd_->g_r_roughness ? sd_->g_r_roughness[r] : 0.0f;
      for (int i = 0; i < N; ++i) {
        float g = (dot3(g3[0], so[i].rgb) + dot3(g3[1], so[i].diffuse)) + dot3(g3[2], so[i].specular);
        if (acc < 1.0f) g -= gsum;
        g += g_acc + g_dist * (0.5f * (td[i] + td[i + 1]));
        g += (dot3(g_rn, so[i].normals) + dot3(g_rnp, so[i].normals_pred)) + (dot3(g_rt, so[i].tint) + g_rr * so[i].roughness);
        if (sd_->g_weights) g += sd_->g_weights[(size_t)r * N + i];
        gw[i] = g;
      }
      float norm_d = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
      double suffix = 0.0, cum = 0.0;
      for (int i = 0; i < N; ++i) cum += (double)(dens[i] * ((td[i + 1] - td[i]) * norm_d));
      for (int i = N - 1; i >= 0; --i) {
        const size_t si = (size_t)r * N + i;
        float delta = (td[i + 1] - td[i]) * norm_d;
        float dd = dens[i] * delta;
        cum -= (double)dd;
        float g_dd = gw[i] * expf(-dd) * expf(-(float)cum) - (float)suffix;
        suffix += (double)(gw[i] * wts[i]);
        float g_density = g_dd * delta;
        if (cfg->opaque_background && i == N - 1) g_density = 0.0f;
        if (sd_->g_density) g_density += sd_->g_density[si];
        float g_rgb_s[3], g_np[3], extra[10];
        for (int c = 0; c < 3; ++c) {
          g_rgb_s[c] = wts[i] * g3[0][c] + (sd_->g_rgb ? sd_->g_rgb[si * 3 + c] : 0.0f);
          g_np[c] = wts[i] * g_rnp[c] + (sd_->g_normals_pred ? sd_->g_normals_pred[si * 3 + c] : 0.0f);
          extra[c] = wts[i] * g3[1][c] + (sd_->g_diffuse ? sd_->g_diffuse[si * 3 + c] : 0.0f);
          extra[3 + c] = wts[i] * g3[2][c] + (sd_->g_specular ? sd_->g_specular[si * 3 + c] : 0.0f);
          extra[6 + c] = wts[i] * g_rt[c] + (sd_->g_tint ? sd_->g_tint[si * 3 + c] : 0.0f);
        }
        extra[9] = wts[i] * g_rr + (sd_->g_roughness ? sd_->g_roughness[si] : 0.0f);
        mlp_backward_x(&model, cfg, &cache[i], v, g_density, g_rgb_s, g_np, extra, G);
      }
    }
#pragma omp critical
    { for (int k = 0; k < model.L.total; ++k) grads[k] += G[k]; }
    free(G); free(cache); free(so); free(buf);
  }
  model_free(&model);
  return 0;
}
