/*
 * refnerf_level_common.h -- phases shared by the fp32 and bf16 level kernels:
 * P0 resample (one wave per ray), P4 per-sample head activations + reflection,
 * P6 colour head + history stores, P7 per-ray alpha scan + compositing.
 * Reference citations inline (file:line relative to the upstream repo root).
 */
#pragma once
#include <hip/hip_runtime.h>

#include "refnerf_hip.h"
#include "refnerf_device_math.h"
#include "refnerf_layout.h"

namespace rn {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

constexpr int NTHREADS = 256;

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4uu __attribute__((ext_vector_type(4)));
/* fp32 pair -> packed bf16 (v_cvt_pk_bf16_f32, emitted by the compiler so that
 * the MFMA-result -> VALU-read wait states are honoured: an inline-asm cvt
 * reading a VGPR accumulator straight after the last MFMA returned garbage);
 * ReLU afterwards on the packed pair as v_pk_max_i16(x, 0): a negative bf16 is
 * a negative int16. */
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  v2bf r = __builtin_convertvector((v2f){lo, hi}, v2bf);
  return __builtin_bit_cast(unsigned, r);
}

constexpr int HD_ROWS = 12;
/* Per-sample floats kept in LDS until the rays of the workgroup are composited:
 * PS[sample][slot] with an odd stride NP (conflict-free per-sample access).
 * NP = NPS_EVAL (17 slots) or NPS_TRAIN (+ the density-gradient normals).  What only
 * the per-sample history needs (grad_pred) lives in a per-pass tile PX[column][3]. */
constexpr int NPS_EVAL = 17, NPS_TRAIN = 21;
enum { PS_DENSITY = 0, PS_RGB = 1, PS_DIF = 4, PS_SPC = 7, PS_NPRED = 10, PS_TINT = 13, PS_ROUGH = 16, PS_NORMALS = 17 };

struct LevelArgs {
  const void *packed;
  refnerf_level_cfg cfg;
  refnerf_rays rays;
  int R;
  int rpw;  /* rays per workgroup */
  const float *sdist_in;
  const float *weights_in;
  refnerf_level_out out;
  long long *prof;   /* debug: per-phase cycle stamps of workgroup 0 (REFNERF_PROF=1), else NULL */
  /* MLP stage entry (refnerf_mlp_forward): caller-supplied Gaussians instead of resample + cast */
  const float *g_means;   /* [R,N,3] */
  const float *g_covs;    /* [R,N,3,3] (cov_full) or [R,N,3] diagonal */
  int cov_full;
  /* training forward: every linear layer's input is saved for the backward as the ACT matrix
   * [ACT_ROWS][act_pitch] (refnerf_layout.h), or NULL */
  float *act;
  long long act_pitch;
  int ring_off;           /* bf16 chains: byte offset of the shared weight-stream ring (RING_BYTES) in dynamic LDS */
};

#define RN_STAMP(A, slot) do { asm volatile("; RNMARK " #slot); if ((A).prof && blockIdx.x == (gridDim.x >> 1) && (threadIdx.x & 63) == 0) (A).prof[(threadIdx.x >> 6) * 32 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

/* Index into an LDS tile of 128-float rows for a compile-time row: rows whose byte offset passes the 64 KB immediate
 * range of the ds instructions go through `hi` = a LAUNDERED 128 * 128 + column (tile_hi()), so that one register
 * serves all of them.  Left to itself the compiler materialises one address register per such row, hoists them in
 * front of the pass loop and spills them: the 73 reads of ide_grad took 44 k cycles of scratch reloads (2.4 k now). */
__device__ __forceinline__ int tile_hi(int col) {
  int hi = 128 * 128 + col;
  asm volatile("" : "+v"(hi));
  return hi;
}
__device__ __forceinline__ int tile_idx(int row, int col, int hi) { return row < 128 ? row * 128 + col : (row - 128) * 128 + hi; }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
/* Sum K per-lane values over the wave through an LDS transpose: lane k adds
 * the 64 partials of value k (65-float pitch: conflict-free both ways), then
 * every total is broadcast.  ~K/6 of the instructions (and time) of K butterfly
 * reductions.  `scr` = 65*K floats of wave-private LDS. */
constexpr int WSUM_PITCH = 65;
template <int K>
__device__ __forceinline__ void wave_sum_many(float (&v)[K], float *scr, int lane) {
#pragma unroll
  for (int k = 0; k < K; ++k) scr[k * WSUM_PITCH + lane] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float s = 0.0f;
  if (lane < K) {
    const float *col = scr + lane * WSUM_PITCH;
#pragma clang loop unroll(disable)
    for (int j = 0; j < 64; j += 4) s += (col[j] + col[j + 1]) + (col[j + 2] + col[j + 3]);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __shfl(s, k, 64);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

/* value of lane `l` (compile-time) as a wave-uniform scalar: v_readlane_b32 */
__device__ __forceinline__ float lane_value(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

/* inclusive scan of doubles across the wave */
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

/* stepfun.sample_intervals (stepfun.py:209-258) for one ray, executed by one
 * wave.  t_in[M+1], logits in LDS scratch `lg[M]`; writes sdist[N+1] to `sd`
 * (LDS) and optional bin indices.  Scratch: e[M] (aliases lg), cw[M+1], c[N].
 * The softmax sum and the float64 cumsum are SEQUENTIAL chains in the oracle's / torch's accumulation order, so that the
 * CDF is bit-identical to theirs; the chain runs wave-uniformly on values fetched with v_readlane (no LDS round trip per
 * term), every lane ends up with the same sums. */
template <bool EXACT = true>
__device__ __forceinline__ void sample_intervals_wave(const float *t_in, float *lg, float *cw, float *c, int M, int N,
                                      float smin, float smax, float *sd, int32_t *bin_idx_g, int lane) {
  /* softmax: max is order-independent */
  float mx = -INFINITY;
  #pragma clang loop unroll(disable)
  for (int i = lane; i < M; i += 64) mx = fmaxf(mx, lg[i]);
  mx = wave_max(mx);
  #pragma clang loop unroll(disable)
  for (int i = lane; i < M; i += 64) lg[i] = rn_det_expf(lg[i] - mx);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float sum = 0.0f;
  if (EXACT) {
    /* the sequential fp32 sum e_0 + e_1 + ... (torch's order), evaluated by EVERY lane on wave-uniform values: 64
     * elements per trip travel lane -> SGPR (v_readlane) instead of through 64 dependent LDS round trips on lane 0
     * (13 k -> 1.5 k cycles at M = 128; elements past M are +0.0, which leaves an fp32 sum unchanged) */
    #pragma clang loop unroll(disable)
    for (int i0 = 0; i0 < M; i0 += 64) {
      const float x = (i0 + lane < M) ? lg[i0 + lane] : 0.0f;
#pragma unroll
      for (int l = 0; l < 64; ++l) sum += lane_value(x, l);
    }
  } else {
    #pragma clang loop unroll(disable)
    for (int i = lane; i < M; i += 64) sum += lg[i];
    sum = wave_sum(sum);
  }
  #pragma clang loop unroll(disable)
  for (int i = lane; i < M; i += 64) lg[i] = lg[i] / sum;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (EXACT) {
    /* float64 running sum of p_0 .. p_{M-2} in index order (torch.cumsum on the reference's float64 promotion), again as a
     * wave-uniform chain over v_readlane values; lane l keeps the partial sum that ends at its own element */
    double acc = 0.0;
    #pragma clang loop unroll(disable)
    for (int i0 = 0; i0 < M; i0 += 64) {
      const float x = (i0 + lane < M) ? lg[i0 + lane] : 0.0f;
      float mine = 0.0f;
#pragma unroll
      for (int l = 0; l < 64; ++l) {
        acc += (double)lane_value(x, l);
        mine = (lane == l) ? (float)acc : mine;
      }
      if (i0 + lane < M - 1) cw[i0 + lane + 1] = fminf(1.0f, mine);
    }
    if (lane == 0) { cw[0] = 0.0f; cw[M] = 1.0f; }
  } else {
    /* wave-parallel prefix sum (float64 partials, chunk per lane): same CDF up to
     * the summation order, not bit-identical to the sequential one */
    const int Cn = (M + 63) / 64, i0 = lane * Cn;
    double loc = 0.0;
    #pragma clang loop unroll(disable)
    for (int q = 0; q < Cn; ++q) if (i0 + q < M - 1) loc += (double)lg[i0 + q];
    double run = wave_scan_incl(loc, lane) - loc;
    #pragma clang loop unroll(disable)
    for (int q = 0; q < Cn; ++q) if (i0 + q < M - 1) { run += (double)lg[i0 + q]; cw[i0 + q + 1] = fminf(1.0f, (float)run); }
    if (lane == 0) { cw[0] = 0.0f; cw[M] = 1.0f; }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  /* inverse CDF at the deterministic centres (math.py:88-111) */
  #pragma clang loop unroll(disable)
  for (int k = lane; k < N; k += 64) {
    float u = linspace_u(k, N);
    /* lo = max{j : u >= cw[j]}; cw is non-decreasing, cw[0]=0 <= u < 1=cw[M] */
    int lo = 0, hi = M;  /* invariant: cw[lo] <= u, cw[hi] > u */
    #pragma clang loop unroll(disable)
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (u >= cw[mid]) lo = mid; else hi = mid;
    }
    float xp0 = cw[lo], xp1 = cw[lo + 1], fp0 = t_in[lo], fp1 = t_in[lo + 1];
    float q = (u - xp0) / (xp1 - xp0);
    if (q != q) q = 0.0f;                       /* nan_to_num(., 0) */
    float off = clip01(q);                      /* +-inf clip like +-FLT_MAX */
    c[k] = fp0 + off * (fp1 - fp0);
    if (bin_idx_g) bin_idx_g[k] = lo;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  #pragma clang loop unroll(disable)
  for (int k = lane; k <= N; k += 64) {
    float v;
    if (k == 0) v = fmaxf(smin, 2.0f * c[0] - (c[1] + c[0]) / 2.0f);
    else if (k == N) v = fminf(smax, 2.0f * c[N - 1] - (c[N - 1] + c[N - 2]) / 2.0f);
    else v = (c[k] + c[k - 1]) / 2.0f;
    sd[k] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

/* this lane's index within the wave, formed HERE (round 6): a lane constant derived from it lives from this point on, not from
 * the kernel's entry across every trunk in front of it (in scratch, once the kernel sits at its register limit) */
__device__ __forceinline__ int fresh_lane() {
  /* The lane index from the mbcnt BUILTINS on an opaque zero: not hoistable, not merged with the kernel's entry value -- and made
   * of instructions the compiler knows.  Round 6 first had the two v_mbcnt in an `asm volatile`: the hazard recogniser does not
   * look inside inline asm, and where the allocator gave the asm's output a register of a just-issued MFMA's dead accumulator
   * (the rgb slice: 13 of its 16 result registers are never read) the MFMA's late write-back raced the v_mbcnt -- a lane index
   * that was sometimes an accumulator value (found by bisecting a rewrite of the training forward's P6 that failed from run to
   * run: docs/EXPERIMENTS.md section 11).  No inline asm in this tree holds a vector instruction any more. */
  int z = 0;
  asm volatile("" : "+v"(z));
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
}

__device__ __forceinline__ void st3(float *base, size_t idx, float a, float b, float c) {
  if (base) { base[idx * 3 + 0] = a; base[idx * 3 + 1] = b; base[idx * 3 + 2] = c; }
}


/* P0 (models.py:200-218): resample every ray of the workgroup, one wave per
 * ray; writes metric distances tdist to TD[rl][N+1] (LDS) and sdist / bin
 * indices to HBM.  `scratch` needs min(rpw,4) * (3*(M+4) + N+3) floats. */
template <int NW = 4, bool EXACT = true>
__device__ __forceinline__ void resample_phase(const LevelArgs &A, float *scratch, float *TD, float *NRM, int ray0, int wave, int lane) {
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples, M = cfg.n_in, rpw = A.rpw;
  const int Mp = (M + 4) & ~3, Np = (N + 3) & ~3;      /* per-wave scratch: 3*Mp + Np floats */
  float *scr = scratch + wave * (3 * Mp + Np);
  float *t_in = scr, *lg = scr + Mp, *cw = scr + 2 * Mp, *c = scr + 3 * Mp;
  #pragma clang loop unroll(disable)
  for (int rl = wave; rl < rpw; rl += NW) {
    int ray = ray0 + rl;
    if (ray >= A.R) break;
    const float *tg = A.sdist_in + (size_t)ray * (M + 1);
    const float *wg = A.weights_in + (size_t)ray * M;
    #pragma clang loop unroll(disable)
    for (int i = lane; i <= M; i += 64) t_in[i] = tg[i];
    wave_sync();
    /* models.py:200-203 */
    #pragma clang loop unroll(disable)
    for (int i = lane; i < M; i += 64)
#ifdef REFNERF_EXP_LOGF_OCML      /* A/B only: the device libm's logf (rounds 1-5) */
      lg[i] = (t_in[i + 1] > t_in[i]) ? cfg.anneal * logf(wg[i] + cfg.resample_padding) : -INFINITY;
#else
      lg[i] = (t_in[i + 1] > t_in[i]) ? cfg.anneal * rn_det_logf(wg[i] + cfg.resample_padding) : -INFINITY;   /* shared with the oracle: bit-identical logits */
#endif
    wave_sync();
    float *sd = TD + rl * (N + 1);
    sample_intervals_wave<EXACT>(t_in, lg, cw, c, M, N, cfg.s_near, cfg.s_far, sd,
                          A.out.d_bin_idx ? A.out.d_bin_idx + (size_t)ray * N : nullptr, lane);
    float nearv = A.rays.d_near[ray], farv = A.rays.d_far[ray];
    /* bf16 kernel only (the fp32 kernels keep their proven code path): the ray's geometry goes to LDS once --
     * NRM[rl] = |d| for the compositing phase and, behind the rpw norms, 12 floats per ray
     * [o(3) d(3) viewdir(3) radius] that the per-sample phases then read as LDS broadcasts instead of
     * 64 identical global loads per phase */
    if (!EXACT) {
      float *RY = NRM + 8 + rl * 12;
      if (lane < 10) {
        const float val = lane < 3 ? A.rays.d_origins[(size_t)ray * 3 + lane]
                        : lane < 6 ? A.rays.d_directions[(size_t)ray * 3 + lane - 3]
                        : lane < 9 ? A.rays.d_viewdirs[(size_t)ray * 3 + lane - 6] : A.rays.d_radii[ray];
        RY[lane] = val;
        const float dx = __shfl(val, 3, 64), dy = __shfl(val, 4, 64), dz = __shfl(val, 5, 64);
        if (lane == 0) NRM[rl] = sqrtf((dx * dx + dy * dy) + dz * dz);
      }
    }
    #pragma clang loop unroll(disable)
    for (int k = lane; k <= N; k += 64) {
      float s = sd[k];
      if (A.out.d_sdist) A.out.d_sdist[(size_t)ray * (N + 1) + k] = s;
      sd[k] = s_to_t(s, nearv, farv, A.cfg.raydist);         /* models.py:218, coord.py:96-98 */
    }
  }
}

/* P4 (models.py:611-683): head activations, predicted normal, reflection. */
struct SampleHeads {
  float density, rough, dot;
  float tint[3], raw_dif[3], npred[3], gp[3], refd[3];
  float normals[3] = {0.0f, 0.0f, 0.0f};   /* density-gradient normals (training forward) */
};
template <bool FAST = false>
__device__ __forceinline__ void sample_heads(const refnerf_level_cfg &cfg, float raw_density, const float gp[3],
                                             float raw_rough, const float raw_dif[3], const float raw_tint[3],
                                             const float v[3], SampleHeads &s) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.gp[i] = gp[i]; s.raw_dif[i] = raw_dif[i]; s.tint[i] = sigmoid_m<FAST>(raw_tint[i]); }
  float n2 = fmaxf((gp[0] * gp[0] + gp[1] * gp[1]) + gp[2] * gp[2], EPS32);
  float nrm = sqrtf(n2);
#pragma unroll
  for (int i = 0; i < 3; ++i) s.npred[i] = -m_div<FAST>(gp[i], nrm);
  s.density = softplus_m<FAST>(raw_density + cfg.density_bias);
  s.rough = softplus_m<FAST>(raw_rough + cfg.roughness_bias);
  float w3[3] = {-v[0], -v[1], -v[2]};
  float dot = (s.npred[0] * w3[0] + s.npred[1] * w3[1]) + s.npred[2] * w3[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) s.refd[i] = (2.0f * dot) * s.npred[i] - w3[i];
  s.dot = (s.npred[0] * v[0] + s.npred[1] * v[1]) + s.npred[2] * v[2];
}

/* P6 (models.py:699-729): colour head; keeps what compositing needs in LDS
 * PS[c][g] (the history is flushed from there by history_flush). */
/* PSM: 0, or (a power of two) - 1: the per-sample records live in a RING of PSM + 1 rows (sample g -> row g & PSM): the
 * 16-bit inference kernel composites every ray as soon as its last sample is in, so a workgroup can own more samples than
 * fit the LDS at once (rays_per_wg * N a multiple of the 256-sample pass for N = 192, 96, 160 ...) */
template <int PSM>
__device__ __forceinline__ int ps_row(int g) { return PSM ? (g & PSM) : g; }

/* FAST_SRGB: the 5/12 power of the sRGB curve through v_log_f32 / v_exp_f32 (2-3 ulp) instead of the library powf (nine
 * calls of ~100 instructions per sample); set by the split-f16 kernel, whose other transcendentals stay libm-accurate */
/* PAD_HERE: rgb_padding is laundered, so that the padding scale is formed here (two VALU instructions) instead of at the kernel's
 * entry, from where it rides in a VGPR -- in scratch -- to this phase (the split-f16 training forward, round 6) */
template <bool FAST = false, int NP = NPS_TRAIN, int PSM = 0, bool FAST_SRGB = FAST, bool PAD_HERE = false>
__device__ __forceinline__ void colour_store(const LevelArgs &A, const SampleHeads &s, const float raw_rgb[3],
                                             float *PS, float *PX, int n_tot, int g_sample, int gcol) {
  const int g = ps_row<PSM>(g_sample);
  const refnerf_level_cfg &cfg = A.cfg;
  float spec_lin[3], dif_lin[3], rgb[3], dif[3], spc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float sg = sigmoid_m<FAST>(cfg.rgb_premultiplier * raw_rgb[i] + cfg.rgb_bias);
    dif_lin[i] = sigmoid_m<FAST>(s.raw_dif[i] - LOG3_F);
    spec_lin[i] = s.tint[i] * sg;
    rgb[i] = spec_lin[i] + dif_lin[i];
  }
  if (cfg.srgb_mapping) {
    if (cfg.srgb_mapping_normalization) {
      float norm = fmaxf(fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]), 1.0f);
#pragma unroll
      for (int i = 0; i < 3; ++i) rgb[i] = m_div<FAST>(rgb[i], norm);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rgb[i] = clip01(linear_to_srgb<FAST_SRGB>(rgb[i]));
      dif[i] = clip01(linear_to_srgb<FAST_SRGB>(dif_lin[i]));
      spc[i] = clip01(linear_to_srgb<FAST_SRGB>(spec_lin[i]));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) { dif[i] = dif_lin[i]; spc[i] = spec_lin[i]; }
  }
  float pad = cfg.rgb_padding;
  if (PAD_HERE) asm volatile("" : "+s"(pad));
  const float pad_scale = (float)(1.0 + 2.0 * (double)pad);
#pragma unroll
  for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] * pad_scale - pad;
  PS[g * NP + PS_DENSITY] = s.density;
  PS[g * NP + PS_ROUGH] = s.rough;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    PS[g * NP + PS_RGB + i] = rgb[i];
    PS[g * NP + PS_DIF + i] = dif[i];
    PS[g * NP + PS_SPC + i] = spc[i];
    PS[g * NP + PS_NPRED + i] = s.npred[i];
    PS[g * NP + PS_TINT + i] = s.tint[i];
    if (NP > PS_NORMALS) PS[g * NP + PS_NORMALS + i] = s.normals[i];
    PX[gcol * 3 + i] = s.gp[i];
  }
}

/* the per-sample history (84 B/sample, 48 MB per launch at C2) is written once and read by a later kernel:
 * non-temporal, so that it does not push the weight image out of the L2s it is streamed from */
#ifndef REFNERF_HIST_NT
#define REFNERF_HIST_NT 1
#endif
__device__ __forceinline__ void hist_store(float *p, float v) {
#if REFNERF_HIST_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

/* Per-sample history (models.py:731-750) of one wave's 32-sample block, written
 * from LDS PS with fully coalesced stores: for the [R,N,3] tensors lane L writes
 * flat element 3*sample + channel = L, L+64.  gw0 = first sample (workgroup
 * index) of the block, gs0 = its global sample index ray*N + i. */
template <int NP = NPS_TRAIN, int PSM = 0>
__device__ __forceinline__ void history_flush(const LevelArgs &A, const float *PS, const float *PX, int n_tot, int gw0, int gcol0,
                                              size_t gs0, int lane) {
  const size_t total = (size_t)A.R * A.cfg.n_samples;
  auto vec3 = [&](float *dst, int slot) {
    if (!dst) return;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int f = lane + 64 * it;
      const int smp = f / 3, c = f - 3 * smp;
      if (f < 96 && gw0 + smp < n_tot && gs0 + smp < total) hist_store(dst + gs0 * 3 + f, PS[ps_row<PSM>(gw0 + smp) * NP + slot + c]);
    }
  };
  auto scal = [&](float *dst, int slot) {
    if (dst && lane < 32 && gw0 + lane < n_tot && gs0 + lane < total) hist_store(dst + gs0 + lane, PS[ps_row<PSM>(gw0 + lane) * NP + slot]);
  };
  scal(A.out.d_density, PS_DENSITY);
  scal(A.out.d_roughness, PS_ROUGH);
  vec3(A.out.d_rgb, PS_RGB);
  vec3(A.out.d_diffuse, PS_DIF);
  vec3(A.out.d_specular, PS_SPC);
  vec3(A.out.d_normals_pred, PS_NPRED);
  if (A.out.d_grad_pred) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int f = lane + 64 * it;
      const int smp = f / 3;
      if (f < 96 && gw0 + smp < n_tot && gs0 + smp < total) hist_store(A.out.d_grad_pred + gs0 * 3 + f, PX[gcol0 * 3 + f]);
    }
  }
  vec3(A.out.d_tint, PS_TINT);
  if (NP > PS_NORMALS && A.cfg.training) vec3(A.out.d_normals, PS_NORMALS);
}

/* P7: alpha weights + compositing, one wave per ray (render.py:132-149, 152-254). */
/* rays [rl_begin, rl_end) of the workgroup (default: all of them); PSM: ring of per-sample records (see ps_row), in which
 * case the cross-lane sums go through shuffles instead of the LDS scratch `wscr` (it is not free while passes are running) */
template <int NW = 4, bool FAST = false, int NP = NPS_TRAIN, int PSM = 0>
__device__ __forceinline__ void composite_phase(const LevelArgs &A, const float *TD, float *XP, float *PS, int n_tot,
                                                int ray0, int wave, int lane, float *wscr, const float *NRM,
                                                int rl_begin = 0, int rl_end = -1) {
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples, rpw = A.rpw;
  if (rl_end < 0) rl_end = rpw;
  #pragma clang loop unroll(disable)
  for (int rl = rl_begin + wave; rl < rl_end; rl += NW) {
    const int ray = ray0 + rl;
    if (ray >= A.R) break;
    const float *td = TD + rl * (N + 1);
    const int base = rl * N;
    const int C = (N + 63) / 64;                 /* samples per lane */
    const int i0 = lane * C;
    float norm;
    if (FAST) norm = NRM[rl];                     /* |d|, parked by the resample phase */
    else {
      float dx = A.rays.d_directions[(size_t)ray * 3], dy = A.rays.d_directions[(size_t)ray * 3 + 1], dz = A.rays.d_directions[(size_t)ray * 3 + 2];
      norm = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
    /* density (slot PS_DENSITY of sample i's record) is overwritten by the weights */
    auto wref = [&](int i) -> float & { return PS[ps_row<PSM>(base + i) * NP + PS_DENSITY]; };
    auto rec = [&](int i, int slot) { return PS[ps_row<PSM>(base + i) * NP + slot]; };
    /* pass 1: local sums of density*delta */
    double local = 0.0;
    #pragma clang loop unroll(disable)
    for (int q = 0; q < C; ++q) {
      int i = i0 + q;
      if (i < N) {
        float dd = wref(i) * ((td[i + 1] - td[i]) * norm);
        /* opaque background (render.py:139-143): the last interval's optical depth is +inf; nothing lies
         * behind it, so it stays out of the prefix sums (inf - inf would poison the exclusive scan) */
        if (!(cfg.opaque_background && i == N - 1)) local += (double)dd;
      }
    }
    double incl = wave_scan_incl(local, lane);
    double cum = incl - local;                   /* exclusive prefix of this lane's chunk */
    /* pass 2: weights + weighted sums */
    float acc = 0, s_rgb[3] = {0, 0, 0}, s_dif[3] = {0, 0, 0}, s_spc[3] = {0, 0, 0}, s_np[3] = {0, 0, 0}, s_tn[3] = {0, 0, 0};
    float s_nm[3] = {0, 0, 0};
    float s_dist = 0, s_rgh = 0, s_logd = 0;
    double wlocal = 0.0;
    #pragma clang loop unroll(disable)
    for (int q = 0; q < C; ++q) {
      int i = i0 + q;
      if (i < N) {
        float dd = wref(i) * ((td[i + 1] - td[i]) * norm);
        if (cfg.opaque_background && i == N - 1) dd = INFINITY;
        float alpha = 1.0f - m_exp<FAST>(-dd);
        float trans = m_exp<FAST>(-(float)cum);
        float w = alpha * trans;
        cum += (double)dd;
        wref(i) = w;
        wlocal += (double)w;
        if (A.out.d_weights) A.out.d_weights[(size_t)ray * N + i] = w;
        acc += w;
        float tmid = 0.5f * (td[i] + td[i + 1]);
        s_dist += w * tmid;
        s_logd += w * m_log<FAST>(tmid);
        s_rgh += w * rec(i, PS_ROUGH);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s_rgb[c] += w * rec(i, PS_RGB + c);
          s_dif[c] += w * rec(i, PS_DIF + c);
          s_spc[c] += w * rec(i, PS_SPC + c);
          s_np[c] += w * rec(i, PS_NPRED + c);
          s_tn[c] += w * rec(i, PS_TINT + c);
          if (NP > PS_NORMALS) s_nm[c] += w * rec(i, PS_NORMALS + c);
        }
      }
    }
    {
      float red[22] = {acc, s_dist, s_logd, s_rgh, s_rgb[0], s_rgb[1], s_rgb[2], s_dif[0], s_dif[1], s_dif[2],
                       s_spc[0], s_spc[1], s_spc[2], s_np[0], s_np[1], s_np[2], s_tn[0], s_tn[1], s_tn[2],
                       s_nm[0], s_nm[1], s_nm[2]};
      if constexpr (PSM != 0) {
#pragma unroll
        for (int k = 0; k < 22; ++k) red[k] = wave_sum(red[k]);
      } else wave_sum_many<22>(red, wscr + wave * (22 * WSUM_PITCH), lane);
      acc = red[0]; s_dist = red[1]; s_logd = red[2]; s_rgh = red[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        s_rgb[c] = red[4 + c]; s_dif[c] = red[7 + c]; s_spc[c] = red[10 + c];
        s_np[c] = red[13 + c]; s_tn[c] = red[16 + c]; s_nm[c] = red[19 + c];
      }
    }
    const float bg_w = fmaxf(0.0f, 1.0f - acc);
    float rgb[3], dif[3], spc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rgb[c] = s_rgb[c] + bg_w * cfg.bg_rgb; dif[c] = s_dif[c] + bg_w * cfg.bg_rgb; spc[c] = s_spc[c] + bg_w * cfg.bg_rgb;
    }
    const int mode = cfg.render_srgb_mode;
    if (mode != REFNERF_SRGB_NONE) {              /* render.py:186-216 */
      if (mode == REFNERF_SRGB_NORM_LINEAR || mode == REFNERF_SRGB_NORM_SRGB) {
        float nr = fmaxf(fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]), 1.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] / nr;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (mode == REFNERF_SRGB_SRGB || mode == REFNERF_SRGB_NORM_SRGB) {
          rgb[c] = linear_to_srgb<FAST>(rgb[c]); dif[c] = linear_to_srgb<FAST>(dif[c]); spc[c] = linear_to_srgb<FAST>(spc[c]);
        }
        rgb[c] = clip01(rgb[c]); dif[c] = clip01(dif[c]); spc[c] = clip01(spc[c]);
      }
    }
    if (lane == 0) {
      st3(A.out.d_r_rgb, ray, rgb[0], rgb[1], rgb[2]);
      st3(A.out.d_r_diffuse, ray, dif[0], dif[1], dif[2]);
      st3(A.out.d_r_specular, ray, spc[0], spc[1], spc[2]);
      if (A.out.d_r_distance) A.out.d_r_distance[ray] = s_dist;
      if (A.out.d_r_acc) A.out.d_r_acc[ray] = acc;
      if (cfg.compute_extras) {
        if (cfg.training) {
          if (NP > PS_NORMALS) st3(A.out.d_r_normals, ray, s_nm[0], s_nm[1], s_nm[2]);
          else {
            /* (a kernel without the normals' record columns writes zeros: formed HERE -- as a constant triple the compiler
             * materialises them in front of the pass loop and carries them through it, in scratch in the ring variants) */
            float z = 0.0f;
            asm volatile("" : "+v"(z));
            st3(A.out.d_r_normals, ray, z, z, z);
          }
        }
        st3(A.out.d_r_normals_pred, ray, s_np[0], s_np[1], s_np[2]);
        st3(A.out.d_r_tint, ray, s_tn[0], s_tn[1], s_tn[2]);
        if (A.out.d_r_roughness) A.out.d_r_roughness[ray] = s_rgh;
        if (A.out.d_r_distance_mean) {
          float e = expf(s_logd / fmaxf(EPS32, acc));
          if (e != e) e = INFINITY;
          e = fminf(fmaxf(e, td[0]), td[N]);     /* +-inf -> clip (same as +-FLT_MAX then clip) */
          A.out.d_r_distance_mean[ray] = e;
        }
      }
    }
    /* percentiles (stepfun.py:294-307, math.py:114-142) in float64 */
    if (cfg.compute_extras && A.out.d_r_percentiles) {
      /* knots xp[j], j = 0..N+1: xp[0]=0, xp[j]=min(1,float(cumsum w[0..j-1])), xp[N+1]=1.
       * fp[j] = td[j] (j<=N), fp[N+1]=far. */
      float *xp = XP + rl * (N + 1);
      double wincl = wave_scan_incl(wlocal, lane);
      double run = wincl - wlocal;
      __builtin_amdgcn_wave_barrier();
      #pragma clang loop unroll(disable)
      for (int q = 0; q < C; ++q) {
        int i = i0 + q;
        if (i < N) { run += (double)wref(i); xp[i + 1] = fminf(1.0f, (float)run); }
      }
      if (lane == 0) { xp[0] = 0.0f; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      /* note xp[N] = min(1,cumsum of all N weights) is what integrate_weights
       * produces for index N (weights_aug[:-1] = the N sample weights); xp[N+1] = 1. */
      const float farv = A.rays.d_far[ray];
      const double psd[3] = {(double)(5.0f / 100.0f), (double)(50.0f / 100.0f), (double)(95.0f / 100.0f)};
      const int nk = N + 2;
      if (FAST) {
        /* searchsorted for the three percentiles at once: wave ballots instead of shuffle reductions,
         * then lane p interpolates percentile p */
        int cnt[3] = {0, 0, 0};
        #pragma clang loop unroll(disable)
        for (int j0 = 0; j0 < nk; j0 += 64) {
          const int j = j0 + lane;
          const bool in = j < nk;
          const double xj = (!in || j == nk - 1) ? 1.0 : (double)xp[j];
#pragma unroll
          for (int p = 0; p < 3; ++p) cnt[p] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in && psd[p] >= xj));
        }
        if (lane < 3) {
          const double x = lane == 0 ? psd[0] : (lane == 1 ? psd[1] : psd[2]);
          int idx = (lane == 0 ? cnt[0] : (lane == 1 ? cnt[1] : cnt[2])) - 1;
          if (idx < 0) idx = 0;
          if (idx > nk - 2) idx = nk - 2;
          double x0 = (double)xp[idx], x1 = (idx + 1 == nk - 1) ? 1.0 : (double)xp[idx + 1];
          double f0 = (double)td[idx], f1 = (idx + 1 == nk - 1) ? (double)farv : (double)td[idx + 1];
          double m = (f1 - f0) / (x1 - x0);
          double b = f0 - m * x0;
          A.out.d_r_percentiles[(size_t)ray * 3 + lane] = m * x + b;
        }
      } else {
        #pragma clang loop unroll(disable)
        for (int p = 0; p < 3; ++p) {
          double x = psd[p];
          int cnt = 0;
          #pragma clang loop unroll(disable)
          for (int j = lane; j < nk; j += 64) {
            double xj = (j == nk - 1) ? 1.0 : (double)xp[j];
            cnt += (x >= xj) ? 1 : 0;
          }
#pragma unroll
          for (int o2 = 32; o2 > 0; o2 >>= 1) cnt += __shfl_xor(cnt, o2, 64);
          if (lane == 0) {
            int idx = cnt - 1;
            if (idx < 0) idx = 0;
            if (idx > nk - 2) idx = nk - 2;
            double x0 = (double)xp[idx], x1 = (idx + 1 == nk - 1) ? 1.0 : (double)xp[idx + 1];
            double f0 = (double)td[idx], f1 = (idx + 1 == nk - 1) ? (double)farv : (double)td[idx + 1];
            double m = (f1 - f0) / (x1 - x0);
            double b = f0 - m * x0;
            A.out.d_r_percentiles[(size_t)ray * 3 + p] = m * x + b;
          }
        }
      }
    }
  }
}

}  // namespace rn
