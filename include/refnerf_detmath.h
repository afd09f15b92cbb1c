/*
 * refnerf_detmath.h -- bit-reproducible fp32 helpers shared by the HIP kernels
 * and the CPU oracle.
 *
 * The resampler's CDF-bin index ("sample index") must be
 * bit-exact between implementations on identical (t, logits) inputs.  libm's
 * expf differs by an ulp between glibc, Sleef (torch) and ocml (device), and an
 * ulp in the softmax is enough to flip a bin at a near-tie, so the softmax of
 * stepfun.invert_cdf (stepfun.py:157-165) uses this exp instead: only IEEE
 * fma/mul/add and integer ops, hence identical bits on x86 and gfx950.
 * Accuracy: ~1 ulp on [-86, 0]; inputs below -86 return 0 (softmax terms
 * below 1.6e-38 carry no weight).
 */
#ifndef REFNERF_DETMATH_H
#define REFNERF_DETMATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#define RN_HD __host__ __device__ inline
#else
#include <math.h>
#include <stdint.h>
#include <string.h>
#define RN_HD static inline
#endif

/* exp(x) for x <= 0 (softmax arguments after subtracting the row max). */
RN_HD float rn_det_expf(float x) {
  if (!(x > -86.0f)) return (x != x) ? x : 0.0f;
  if (x > 88.0f) x = 88.0f;
  /* n = round(x / ln2); r = x - n*ln2 in two steps (Cody-Waite) */
  const float LOG2E = 1.44269504088896341f;
  const float LN2_HI = 0.693145751953125f;        /* 0x3f317200 */
  const float LN2_LO = 1.42860682030941723212e-6f;
  float fn = x * LOG2E;
  /* round to nearest, ties away is fine (any consistent choice works) */
  fn = (float)(int)(fn + (fn < 0.0f ? -0.5f : 0.5f));
  float r = fmaf(-fn, LN2_HI, x);
  r = fmaf(-fn, LN2_LO, r);
  /* exp(r), |r| <= ln2/2: degree-6 minimax-ish Taylor/Remez blend */
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float e = fmaf(p, r2, r) + 1.0f;
  /* scale by 2^n through the exponent field; n in [-126, 127] here */
  int n = (int)fn;
  union { float f; unsigned u; } s;
  s.u = (unsigned)(n + 127) << 23;
  return e * s.f;
}

/* log(x), x > 0, for the resampling logits `anneal * log(w + padding)` of models.py:200-203 (round 6).  The logits
 * feed the softmax above: an ulp of difference between glibc's, Sleef's and ocml's logf moves the CDF knots and can
 * flip the bin of a quantile at a near-tie, so kernels and oracle share this one.  Evaluated in float64 with IEEE
 * fma / mul / add and integer operations only (no division, no table): x = 2^e * m, m in [sqrt(1/2), sqrt(2)),
 * s = (m - 1) / (m + 1) through a Newton reciprocal, log m = 2 atanh(s) as an odd series to s^17 (|s| <= 0.1716:
 * truncation 3e-14 relative), result = e * ln2 + log m rounded once to fp32 -- the correctly rounded logf except
 * within ~1e-8 ulp of a rounding boundary, and the same bits on x86 and gfx950. */
RN_HD float rn_det_logf(float x) {
  union { float f; unsigned u; } b;
  b.f = x;
  if (!(x > 0.0f)) {                      /* 0 -> -inf; negative or NaN -> NaN */
    b.u = (x == 0.0f) ? 0xff800000u : 0x7fc00000u;
    return b.f;
  }
  if (b.u >= 0x7f800000u) return x;       /* +inf */
  int e = 0;
  if (b.u < 0x00800000u) { b.f = x * 8388608.0f; e = -23; }   /* subnormal: scale by 2^23 (exact) */
  unsigned ux = b.u + (0x3f800000u - 0x3f3504f3u);
  e += (int)(ux >> 23) - 127;
  b.u = (ux & 0x007fffffu) + 0x3f3504f3u;                      /* m in [sqrt(1/2), sqrt(2)) */
  const double f = (double)b.f - 1.0;                          /* exact */
  const double d = 2.0 + f;                                    /* exact; in [1.707, 2.415) */
  double y = fma(-0.2, d, 0.9);                                /* 1/d within 5 % */
  double r = fma(-d, y, 1.0); y = fma(y, r, y);
  r = fma(-d, y, 1.0); y = fma(y, r, y);
  r = fma(-d, y, 1.0); y = fma(y, r, y);
  r = fma(-d, y, 1.0); y = fma(y, r, y);
  const double s = f * y, s2 = s * s;
  double p = 2.0 / 17.0;
  p = fma(p, s2, 2.0 / 15.0);
  p = fma(p, s2, 2.0 / 13.0);
  p = fma(p, s2, 2.0 / 11.0);
  p = fma(p, s2, 2.0 / 9.0);
  p = fma(p, s2, 2.0 / 7.0);
  p = fma(p, s2, 2.0 / 5.0);
  p = fma(p, s2, 2.0 / 3.0);
  const double lm = fma(s * s2, p, 2.0 * s);
  return (float)fma((double)e, 0.69314718055994530942, lm);
}

#endif
