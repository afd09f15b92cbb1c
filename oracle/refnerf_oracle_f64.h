/*
 * refnerf_oracle_f64.h -- type / libm switch of the float64 "truth" build of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Included by refnerf_oracle.c (after the system headers and refnerf_detmath.h, before
 * refnerf_oracle.h) when compiled with -DRN_ORACLE_F64: every `float` of the restatement -- parameters, rays, step
 * functions, activations, outputs, the cfg struct's real fields -- becomes `double`, every fp32 libm call its
 * float64 twin, and the two bit-reproducible fp32 helpers of refnerf_detmath.h become libm's exp / log.  The
 * algorithm, the operation order and the reference's fp32-derived CONSTANTS (eps = 2^-23, the 100 pi wrap of
 * math.safe_sin, the sRGB knee) are unchanged: the result is the reference's function evaluated with 2^-53 instead
 * of 2^-24 rounding, the quantity both fp32 evaluations (the reference's and the HIP kernels') approximate.
 * Binding: oracle/oracle_f64.py.
 */
#ifndef REFNERF_ORACLE_F64_H
#define REFNERF_ORACLE_F64_H
#define float double
#define expf exp
#define logf log
#define log1pf log1p
#define powf pow
#define sqrtf sqrt
#define sinf sin
#define cosf cos
#define fabsf fabs
#define fmaf fma
#define fmaxf fmax
#define fminf fmin
#define fmodf fmod
#define ldexpf ldexp
#define rn_det_expf exp
#define rn_det_logf log
#undef FLT_MAX
#define FLT_MAX DBL_MAX
#endif
