/*
 * refnerf_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of the Ref-NeRF
 * rendering inner loop.  Written for gfx950 only: 64-wide wavefronts, MFMA,
 * 160 KiB LDS per CU.
 *
 * Kernel map (one launch per sampling level):
 *   level_fwd_f32 : workgroup = 4 waves = RPW whole rays; each wave owns
 *   32-sample blocks.  resample -> warp -> conical frusta -> IPE (LDS) ->
 *   8x256 spatial MLP -> heads -> reflect + IDE (LDS) -> 8x256 directional MLP
 *   -> colour -> per-ray alpha scan + compositing.  The MLP runs transposed,
 *   D[out][sample] = W x X on v_mfma_f32_32x32x2_f32, so a layer's output
 *   registers ARE the next layer's B operands: activations never leave the
 *   register file; only the encodings (IPE 96, dir-MLP input 202) sit in LDS.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "refnerf_hip.h"
#include "refnerf_device_math.h"
#include "refnerf_layout.h"

namespace rn {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int T_TILE = 128;  /* samples per pass: 4 waves x 32 */
constexpr int NTHREADS = 256;
constexpr int HD_ROWS = 12;
constexpr int NPS = 20;      /* per-sample floats kept for compositing */
/* per-sample slots in LDS PS[c][sample] */
enum { PS_DENSITY = 0, PS_RGB = 1, PS_DIF = 4, PS_SPC = 7, PS_NPRED = 10, PS_TINT = 13, PS_ROUGH = 16, PS_NORMALS = 17 };

/* ------------------------------------------------------------------ */
/* weight packing                                                     */
/* ------------------------------------------------------------------ */

/* W[row][k] of GEMM op `op` in canonical storage; k = canonical input column. */
__device__ float canon_w(const float *P, int op, int row, int k) {
  if (op < 8) {
    int in = CANON.sp_in[op];
    return (k < in) ? P[CANON.sp_w[op] + row * in + k] : 0.0f;
  }
  if (op == OP_HEADS) {
    if (row < BNECK) return P[CANON.bneck_w + row * WIDTH + k];
    if (row == HROW_DENSITY) return P[CANON.density_w + k];
    if (row < HROW_ROUGH) return P[CANON.gradpred_w + (row - HROW_GRAD) * WIDTH + k];
    if (row == HROW_ROUGH) return P[CANON.rough_w + k];
    if (row < HROW_TINT) return P[CANON.diffuse_w + (row - HROW_DIFFUSE) * WIDTH + k];
    if (row < HROWS) return P[CANON.tint_w + (row - HROW_TINT) * WIDTH + k];
    return 0.0f;
  }
  if (op < OP_RGB) {
    int i = op - 9, in = CANON.vd_in[i];
    return (k < in) ? P[CANON.vd_w[i] + row * in + k] : 0.0f;
  }
  return (row < 3) ? P[CANON.rgb_w + row * WIDTH + k] : 0.0f;
}
__device__ float canon_b(const float *P, int op, int row) {
  if (op < 8) return P[CANON.sp_b[op] + row];
  if (op == OP_HEADS) {
    if (row < BNECK) return P[CANON.bneck_b + row];
    if (row == HROW_DENSITY) return P[CANON.density_b];
    if (row < HROW_ROUGH) return P[CANON.gradpred_b + row - HROW_GRAD];
    if (row == HROW_ROUGH) return P[CANON.rough_b];
    if (row < HROW_TINT) return P[CANON.diffuse_b + row - HROW_DIFFUSE];
    if (row < HROWS) return P[CANON.tint_b + row - HROW_TINT];
    return 0.0f;
  }
  if (op < OP_RGB) return P[CANON.vd_b[op - 9] + row];
  return (row < 3) ? P[CANON.rgb_b + row] : 0.0f;
}

__global__ void pack_weights_f32(const float *__restrict__ P, float *__restrict__ out) {
  int op = blockIdx.y;
  const Op o = PACKED.op[op];
  int n_a = (o.reg_steps + o.lds_steps) * 64 * o.stride;
  int n_b = o.nob * 32;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_a + n_b; e += gridDim.x * blockDim.x) {
    if (e < n_a) {
      int ob = e % o.stride, lane = (e / o.stride) % 64, step = e / (o.stride * 64);
      int h = lane >> 5, row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (step < o.reg_steps) {
          int kb = step >> 4, r = step & 15;
          int k = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
          v = canon_w(P, op, row, k);
        } else {
          int kl = 2 * (step - o.reg_steps) + h;
          int k = (o.reg_steps ? WIDTH : 0) + kl;
          int valid_k = (op == 0 || op == 5) ? IPE_DIM : DIR_IN;
          v = (kl < valid_k) ? canon_w(P, op, row, k) : 0.0f;
        }
      }
      out[o.a_off + e] = v;
    } else {
      int b = e - n_a;
      int reg = b & 15, h = (b >> 4) & 1, ob = b >> 5;
      int row = ob * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      out[o.b_off + b] = canon_b(P, op, row);
    }
  }
}

/* ------------------------------------------------------------------ */
/* fp32 MFMA GEMM op on one 32-sample block                           */
/* ------------------------------------------------------------------ */

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

/* A fragments come through a buffer descriptor over the packed image: the
 * per-lane part of the address is one constant VGPR (lane*STRIDE*4), the
 * per-step part is an SGPR/immediate, so the 1000+ loads of an op cost no VALU
 * address arithmetic (flat global loads made hipcc precompute and spill
 * hundreds of 64-bit pointers). */
template <int NOB, int STRIDE>
__device__ __forceinline__ void load_a(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float (&a)[NOB]) {
  if constexpr (STRIDE == 8) {
    v4f x = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    a[0] = x[0];
    if constexpr (NOB > 1) { a[1] = x[1]; a[2] = x[2]; a[3] = x[3]; }
    if constexpr (NOB == 5) a[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 16, soff, 0));
    if constexpr (NOB == 8) {
      v4f y = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, soff, 0));
      a[4] = y[0]; a[5] = y[1]; a[6] = y[2]; a[7] = y[3];
    }
  } else {
    a[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
  }
}

/* out[ob] = bias + W * in, W from the packed image `rs`.  a_off/b_off: float
 * offsets of the op inside the image (wave-uniform); `xl` = LDS X + h*T_TILE +
 * column (for LDS steps).  The A stream is software-pipelined PF steps ahead
 * through a register ring; sched_barrier pins "MFMAs of step s, then the loads
 * of step s+PF" so that hipcc cannot sink the loads back to their uses (it
 * otherwise emits load; s_waitcnt vmcnt(0); mfma).  lds_steps % PF == 0. */
constexpr int PF = 3;
template <int NOB, int STRIDE, bool HAS_REG>
__device__ __forceinline__ void gemm_op(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h,
                                        const v16f (&in)[8], v16f (&out)[NOB], const float *xl,
                                        int lds_steps) {
  constexpr int STEP_BYTES = 64 * STRIDE * 4;
  const int voff = lane * STRIDE * 4;
  int soff = a_off * 4;
  float a[PF][NOB];
#pragma unroll
  for (int d = 0; d < PF; ++d) load_a<NOB, STRIDE>(rs, voff, soff + d * STEP_BYTES, a[d]);
  soff += PF * STEP_BYTES;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    v4f b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      b[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, h * 64 + q * 16, b_off * 4 + ob * 128, 0));
    out[ob] = (v16f){b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], b[1][2], b[1][3],
                     b[2][0], b[2][1], b[2][2], b[2][3], b[3][0], b[3][1], b[3][2], b[3][3]};
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (HAS_REG) {
#pragma unroll
    for (int step = 0; step < REG_STEPS; ++step) {
      const float b = in[step >> 4][step & 15];
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
        out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step % PF][ob], b, out[ob], 0, 0, 0);
      load_a<NOB, STRIDE>(rs, voff, soff + step * STEP_BYTES, a[step % PF]);
      __builtin_amdgcn_sched_barrier(0);
    }
    soff += REG_STEPS * STEP_BYTES;
  }
  constexpr int P0 = HAS_REG ? (REG_STEPS % PF) : 0;
  float bcur = xl[0];
#pragma unroll 1
  for (int s = 0; s < lds_steps; s += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const float bnext = xl[2 * (s + u + 1) * T_TILE];
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
        out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(P0 + u) % PF][ob], bcur, out[ob], 0, 0, 0);
      load_a<NOB, STRIDE>(rs, voff, soff + u * STEP_BYTES, a[(P0 + u) % PF]);
      bcur = bnext;
      __builtin_amdgcn_sched_barrier(0);
    }
    soff += PF * STEP_BYTES;
  }
}

__device__ __forceinline__ void relu_into(const v16f (&out)[8], v16f (&in)[8]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) in[ob][r] = fmaxf(out[ob][r], 0.0f);
}

/* ------------------------------------------------------------------ */
/* level kernel                                                       */
/* ------------------------------------------------------------------ */

struct LevelArgs {
  const float *packed;
  refnerf_level_cfg cfg;
  refnerf_rays rays;
  int R;
  int rpw;  /* rays per workgroup */
  const float *sdist_in;
  const float *weights_in;
  refnerf_level_out out;
};

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
 * The softmax sum and the float64 cumsum run sequentially on lane 0 so that the
 * CDF is bit-identical to the oracle / torch's accumulation order. */
__device__ void sample_intervals_wave(const float *t_in, float *lg, float *cw, float *c, int M, int N,
                                      float smin, float smax, float *sd, int32_t *bin_idx_g, int lane) {
  /* softmax: max is order-independent */
  float mx = -INFINITY;
  for (int i = lane; i < M; i += 64) mx = fmaxf(mx, lg[i]);
  mx = wave_max(mx);
  for (int i = lane; i < M; i += 64) lg[i] = rn_det_expf(lg[i] - mx);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float sum = 0.0f;
  if (lane == 0) {
    for (int i = 0; i < M; ++i) sum += lg[i];
  }
  sum = __shfl(sum, 0, 64);
  for (int i = lane; i < M; i += 64) lg[i] = lg[i] / sum;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    cw[0] = 0.0f;
    double acc = 0.0;
    for (int i = 0; i < M - 1; ++i) { acc += (double)lg[i]; cw[i + 1] = fminf(1.0f, (float)acc); }
    cw[M] = 1.0f;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  /* inverse CDF at the deterministic centres (math.py:88-111) */
  for (int k = lane; k < N; k += 64) {
    float u = linspace_u(k, N);
    /* lo = max{j : u >= cw[j]}; cw is non-decreasing, cw[0]=0 <= u < 1=cw[M] */
    int lo = 0, hi = M;  /* invariant: cw[lo] <= u, cw[hi] > u */
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

__device__ __forceinline__ void st3(float *base, size_t idx, float a, float b, float c) {
  if (base) { base[idx * 3 + 0] = a; base[idx * 3 + 1] = b; base[idx * 3 + 2] = c; }
}

__global__ __launch_bounds__(NTHREADS) void level_fwd_f32(const LevelArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples, M = cfg.n_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int rpw = A.rpw;
  const int ray0 = blockIdx.x * rpw;
  const int n_tot = rpw * N;                 /* samples owned by this workgroup */

  float *X = smem;                               /* [DIR_PAD][T_TILE]              */
  float *HD = X + DIR_PAD * T_TILE;              /* [HD_ROWS][T_TILE]              */
  float *TD = HD + HD_ROWS * T_TILE;             /* [rpw][N+1] metric distances    */
  float *XP = TD + rpw * (N + 1);                /* [rpw][N+1] CDF knots for the percentiles */
  float *PS = XP + rpw * (N + 1);                /* [NPS][n_tot]                   */

  /* ---------------- P0: resample (one wave per ray) ---------------- */
  {
    float *scr = X + wave * (3 * 520 + 8);       /* per-wave scratch inside X */
    float *t_in = scr, *lg = scr + 520, *cw = scr + 1040;
    float *c = X + 4 * (3 * 520 + 8) + wave * 648;   /* sample centres, N <= 640 */
    for (int rl = wave; rl < rpw; rl += 4) {
      int ray = ray0 + rl;
      if (ray >= A.R) break;
      const float *tg = A.sdist_in + (size_t)ray * (M + 1);
      const float *wg = A.weights_in + (size_t)ray * M;
      for (int i = lane; i <= M; i += 64) t_in[i] = tg[i];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      /* models.py:200-203 */
      for (int i = lane; i < M; i += 64)
        lg[i] = (t_in[i + 1] > t_in[i]) ? cfg.anneal * logf(wg[i] + cfg.resample_padding) : -INFINITY;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float *sd = TD + rl * (N + 1);
      sample_intervals_wave(t_in, lg, cw, c, M, N, cfg.s_near, cfg.s_far, sd,
                            A.out.d_bin_idx ? A.out.d_bin_idx + (size_t)ray * N : nullptr, lane);
      float nearv = A.rays.d_near[ray], farv = A.rays.d_far[ray];
      for (int k = lane; k <= N; k += 64) {
        float s = sd[k];
        if (A.out.d_sdist) A.out.d_sdist[(size_t)ray * (N + 1) + k] = s;
        sd[k] = s_to_t(s, nearv, farv);         /* models.py:218 */
      }
    }
  }
  __syncthreads();

  /* ---------------- per-pass MLP over 32-sample blocks ---------------- */
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)A.packed, 0, PACKED.total * 4, 0x00020000);
  const int col = wave * 32 + sl;                /* this lane's column in X / HD */
  const float *xl = X + h * T_TILE + col;
  v16f in[8], out[8];

  for (int pass0 = 0; pass0 < n_tot; pass0 += T_TILE) {
    const int g = pass0 + col;                   /* sample index inside the workgroup */
    const int rl = g / N, si = g - rl * N;
    const int ray = ray0 + rl;
    const bool valid = (g < n_tot) && (ray < A.R);
    const int rayc = valid ? ray : (A.R - 1);
    float o[3], d[3], v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = A.rays.d_origins[(size_t)rayc * 3 + i];
      d[i] = A.rays.d_directions[(size_t)rayc * 3 + i];
      v[i] = A.rays.d_viewdirs[(size_t)rayc * 3 + i];
    }
    /* P1: conical frustum -> lifted Gaussian -> IPE (half 0: sin, half 1: cos) */
    {
      float radius = A.rays.d_radii[rayc];
      const float *td = TD + (valid ? rl : 0) * (N + 1);
      float t0 = td[valid ? si : 0], t1 = td[valid ? si + 1 : 1];
      float lm[3], lv[3];
      cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
#pragma unroll 1
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int b = 0; b < 3; ++b) X[(48 * h + j * 3 + b) * T_TILE + col] = ipe_feature(lm[b], lv[b], j, h);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* P2: spatial MLP (models.py:576-580) */
    gemm_op<8, 8, false>(rs, PACKED.op[0].a_off, PACKED.op[0].b_off, lane, h, in, out, xl, PACKED.op[0].lds_steps);
    relu_into(out, in);
#pragma unroll 1
    for (int op = 1; op < 8; ++op) {
      gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps);
      relu_into(out, in);
    }
    /* P3: heads (models.py:582,613,634-645): 4 bottleneck blocks + 1 scalar block */
    {
      v16f hd[5];
      gemm_op<5, 8, true>(rs, PACKED.op[OP_HEADS].a_off, PACKED.op[OP_HEADS].b_off, lane, h, in, hd, xl, 0);
      __builtin_amdgcn_wave_barrier();          /* all IPE reads of this wave are done */
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * T_TILE + col] = hd[blk][r];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < HD_ROWS) HD[row * T_TILE + col] = hd[4][r];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* P4: activations, reflection, IDE (models.py:611-686) */
    float tint[3], raw_dif[3], npred[3], gp[3], density, rough;
    {
      float raw_density = HD[0 * T_TILE + col];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        gp[i] = HD[(1 + i) * T_TILE + col];
        raw_dif[i] = HD[(5 + i) * T_TILE + col];
        tint[i] = sigmoid_t(HD[(8 + i) * T_TILE + col]);
      }
      float raw_rough = HD[4 * T_TILE + col];
      float n2 = fmaxf((gp[0] * gp[0] + gp[1] * gp[1]) + gp[2] * gp[2], EPS32);
      float nrm = sqrtf(n2);
#pragma unroll
      for (int i = 0; i < 3; ++i) npred[i] = -(gp[i] / nrm);
      density = softplus_t(raw_density + cfg.density_bias);
      rough = softplus_t(raw_rough + cfg.roughness_bias);
      float w3[3] = {-v[0], -v[1], -v[2]};
      float dot = (npred[0] * w3[0] + npred[1] * w3[1]) + npred[2] * w3[2];
      float refd[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) refd[i] = (2.0f * dot) * npred[i] - w3[i];
      float *xi = X + (BNECK + IDE_TERMS * h) * T_TILE + col;
      ide_eval(refd[0], refd[1], refd[2], rough, h, [&](int q, float val) { xi[q * T_TILE] = val; });
      if (h == 0) X[(BNECK + IDE_DIM) * T_TILE + col] = (npred[0] * v[0] + npred[1] * v[1]) + npred[2] * v[2];
      else {
#pragma unroll
        for (int q = DIR_IN; q < DIR_PAD; ++q) X[q * T_TILE + col] = 0.0f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* P5: directional MLP (models.py:690-694) + rgb (699-700) */
    gemm_op<8, 8, false>(rs, PACKED.op[9].a_off, PACKED.op[9].b_off, lane, h, in, out, xl, PACKED.op[9].lds_steps);
    relu_into(out, in);
#pragma unroll 1
    for (int op = 10; op < 17; ++op) {
      gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps);
      relu_into(out, in);
    }
    v16f rgbv[1];
    gemm_op<1, 1, true>(rs, PACKED.op[OP_RGB].a_off, PACKED.op[OP_RGB].b_off, lane, h, in, rgbv, xl, 0);
    /* rows 0..2 live in half 0, regs 0..2; hand them to half 1 as well */
    float raw_rgb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(rgbv[0][i], sl, 64);

    /* P6: colour head (models.py:699-729) */
    if (valid && h == 0) {
      float spec_lin[3], dif_lin[3], rgb[3], dif[3], spc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float sg = sigmoid_t(cfg.rgb_premultiplier * raw_rgb[i] + cfg.rgb_bias);
        dif_lin[i] = sigmoid_t(raw_dif[i] - LOG3_F);
        spec_lin[i] = tint[i] * sg;
        rgb[i] = spec_lin[i] + dif_lin[i];
      }
      if (cfg.srgb_mapping) {
        if (cfg.srgb_mapping_normalization) {
          float norm = fmaxf(fmaxf(fmaxf(rgb[0], rgb[1]), rgb[2]), 1.0f);
#pragma unroll
          for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] / norm;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          rgb[i] = clip01(linear_to_srgb(rgb[i]));
          dif[i] = clip01(linear_to_srgb(dif_lin[i]));
          spc[i] = clip01(linear_to_srgb(spec_lin[i]));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) { dif[i] = dif_lin[i]; spc[i] = spec_lin[i]; }
      }
      const float pad_scale = (float)(1.0 + 2.0 * (double)cfg.rgb_padding);
#pragma unroll
      for (int i = 0; i < 3; ++i) rgb[i] = rgb[i] * pad_scale - cfg.rgb_padding;
      PS[PS_DENSITY * n_tot + g] = density;
      PS[PS_ROUGH * n_tot + g] = rough;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        PS[(PS_RGB + i) * n_tot + g] = rgb[i];
        PS[(PS_DIF + i) * n_tot + g] = dif[i];
        PS[(PS_SPC + i) * n_tot + g] = spc[i];
        PS[(PS_NPRED + i) * n_tot + g] = npred[i];
        PS[(PS_TINT + i) * n_tot + g] = tint[i];
        PS[(PS_NORMALS + i) * n_tot + g] = 0.0f;
      }
      size_t gi = (size_t)ray * N + si;
      if (A.out.d_density) A.out.d_density[gi] = density;
      if (A.out.d_roughness) A.out.d_roughness[gi] = rough;
      st3(A.out.d_rgb, gi, rgb[0], rgb[1], rgb[2]);
      st3(A.out.d_diffuse, gi, dif[0], dif[1], dif[2]);
      st3(A.out.d_specular, gi, spc[0], spc[1], spc[2]);
      st3(A.out.d_normals_pred, gi, npred[0], npred[1], npred[2]);
      st3(A.out.d_grad_pred, gi, gp[0], gp[1], gp[2]);
      st3(A.out.d_tint, gi, tint[0], tint[1], tint[2]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  /* ---------------- P7: alpha weights + compositing, one wave per ray ----------------
   * render.py:132-149 and 152-254.  Each lane owns a contiguous chunk of
   * samples so the exclusive float64 cumsum is a local prefix + one wave scan. */
  for (int rl = wave; rl < rpw; rl += 4) {
    const int ray = ray0 + rl;
    if (ray >= A.R) break;
    const float *td = TD + rl * (N + 1);
    const int base = rl * N;
    const int C = (N + 63) / 64;                 /* samples per lane */
    const int i0 = lane * C;
    float dx = A.rays.d_directions[(size_t)ray * 3], dy = A.rays.d_directions[(size_t)ray * 3 + 1], dz = A.rays.d_directions[(size_t)ray * 3 + 2];
    const float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
    float *wbuf = PS + PS_DENSITY * n_tot + base;  /* density is overwritten by the weights */
    /* pass 1: local sums of density*delta */
    double local = 0.0;
    for (int q = 0; q < C; ++q) {
      int i = i0 + q;
      if (i < N) {
        float dd = wbuf[i] * ((td[i + 1] - td[i]) * norm);
        if (cfg.opaque_background && i == N - 1) dd = INFINITY;
        local += (double)dd;
      }
    }
    double incl = wave_scan_incl(local, lane);
    double cum = incl - local;                   /* exclusive prefix of this lane's chunk */
    /* pass 2: weights + weighted sums */
    float acc = 0, s_rgb[3] = {0, 0, 0}, s_dif[3] = {0, 0, 0}, s_spc[3] = {0, 0, 0}, s_np[3] = {0, 0, 0}, s_tn[3] = {0, 0, 0};
    float s_nm[3] = {0, 0, 0};
    float s_dist = 0, s_rgh = 0, s_logd = 0;
    double wlocal = 0.0;
    for (int q = 0; q < C; ++q) {
      int i = i0 + q;
      if (i < N) {
        float dd = wbuf[i] * ((td[i + 1] - td[i]) * norm);
        if (cfg.opaque_background && i == N - 1) dd = INFINITY;
        float alpha = 1.0f - expf(-dd);
        float trans = expf(-(float)cum);
        float w = alpha * trans;
        cum += (double)dd;
        wbuf[i] = w;
        wlocal += (double)w;
        if (A.out.d_weights) A.out.d_weights[(size_t)ray * N + i] = w;
        acc += w;
        float tmid = 0.5f * (td[i] + td[i + 1]);
        s_dist += w * tmid;
        s_logd += w * logf(tmid);
        s_rgh += w * PS[PS_ROUGH * n_tot + base + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s_rgb[c] += w * PS[(PS_RGB + c) * n_tot + base + i];
          s_dif[c] += w * PS[(PS_DIF + c) * n_tot + base + i];
          s_spc[c] += w * PS[(PS_SPC + c) * n_tot + base + i];
          s_np[c] += w * PS[(PS_NPRED + c) * n_tot + base + i];
          s_tn[c] += w * PS[(PS_TINT + c) * n_tot + base + i];
          s_nm[c] += w * PS[(PS_NORMALS + c) * n_tot + base + i];
        }
      }
    }
    acc = wave_sum(acc); s_dist = wave_sum(s_dist); s_logd = wave_sum(s_logd); s_rgh = wave_sum(s_rgh);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_rgb[c] = wave_sum(s_rgb[c]); s_dif[c] = wave_sum(s_dif[c]); s_spc[c] = wave_sum(s_spc[c]);
      s_np[c] = wave_sum(s_np[c]); s_tn[c] = wave_sum(s_tn[c]); s_nm[c] = wave_sum(s_nm[c]);
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
          rgb[c] = linear_to_srgb(rgb[c]); dif[c] = linear_to_srgb(dif[c]); spc[c] = linear_to_srgb(spc[c]);
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
        if (cfg.training) st3(A.out.d_r_normals, ray, s_nm[0], s_nm[1], s_nm[2]);
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
      for (int q = 0; q < C; ++q) {
        int i = i0 + q;
        if (i < N) { run += (double)wbuf[i]; xp[i + 1] = fminf(1.0f, (float)run); }
      }
      if (lane == 0) { xp[0] = 0.0f; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      /* note xp[N] = min(1,cumsum of all N weights) is what integrate_weights
       * produces for index N (weights_aug[:-1] = the N sample weights); xp[N+1] = 1. */
      const float farv = A.rays.d_far[ray];
      const float psf[3] = {5.0f / 100.0f, 50.0f / 100.0f, 95.0f / 100.0f};
      const int nk = N + 2;
      for (int p = 0; p < 3; ++p) {
        double x = (double)psf[p];
        int cnt = 0;
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

/* ------------------------------------------------------------------ */
/* stage kernels                                                      */
/* ------------------------------------------------------------------ */

__global__ __launch_bounds__(256) void sample_intervals_kernel(const float *t, const float *logits, int R, int M, int N,
                                                               float smin, float smax, float *sdist, int32_t *bin_idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (M + 1) + M + (M + 1) + N + (N + 1) + 8;
  float *scr = smem + wave * per;
  float *t_in = scr, *lg = t_in + (M + 1), *cw = lg + M, *c = cw + (M + 1), *sd = c + N;
  int ray = blockIdx.x * 4 + wave;
  if (ray >= R) return;
  for (int i = lane; i <= M; i += 64) t_in[i] = t[(size_t)ray * (M + 1) + i];
  for (int i = lane; i < M; i += 64) lg[i] = logits[(size_t)ray * M + i];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  sample_intervals_wave(t_in, lg, cw, c, M, N, smin, smax, sd, bin_idx ? bin_idx + (size_t)ray * N : nullptr, lane);
  for (int k = lane; k <= N; k += 64) sdist[(size_t)ray * (N + 1) + k] = sd[k];
}

__global__ void ipe_kernel(const float *lmean, const float *lvar, int n, float *feat) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lm[3] = {lmean[i * 3], lmean[i * 3 + 1], lmean[i * 3 + 2]};
  float lv[3] = {lvar[i * 3], lvar[i * 3 + 1], lvar[i * 3 + 2]};
  for (int hb = 0; hb < 2; ++hb)
    for (int j = 0; j < 16; ++j)
      for (int b = 0; b < 3; ++b) feat[(size_t)i * IPE_DIM + 48 * hb + j * 3 + b] = ipe_feature(lm[b], lv[b], j, hb);
}

__global__ void ide_kernel(const float *xyz, const float *kappa_inv, int n, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float *o = out + (size_t)i * IDE_DIM;
  for (int part = 0; part < 2; ++part)
    ide_eval(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], kappa_inv[i], part,
             [&](int q, float val) { o[part * IDE_TERMS + q] = val; });
}

}  // namespace rn

/* ================================================================== */
/* C ABI                                                              */
/* ================================================================== */

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char *fmt, const char *detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}
#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) return fail(REFNERF_EHIP, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

std::once_flag g_tab_once;
int g_tab_status = 0;
/* kernel timing: event pairs recorded on the launch stream, resolved lazily in
 * refnerf_get_timing() so the timed region is not serialised by event syncs */
bool g_timing = false;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events;
size_t g_events_used = 0;

double fact(int n) { double r = 1; for (int i = 2; i <= n; ++i) r *= i; return r; }

void upload_tables() {
  float c[17], a[17 * 17], b[17 * 17];
  for (int m = 0; m <= 16; ++m) {
    double df = 1;
    for (int i = 1; i <= m; ++i) df *= (2 * i - 1);
    double cm = ((m & 1) ? -1.0 : 1.0) * std::sqrt((2.0 * m + 1.0) / (4.0 * M_PI * fact(2 * m))) * df;
    c[m] = (float)cm;
    for (int l = 0; l <= 16; ++l) {
      a[m * 17 + l] = 0; b[m * 17 + l] = 0;
      if (l > m) {
        a[m * 17 + l] = (float)std::sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
        if (l > m + 1) b[m * 17 + l] = (float)std::sqrt((((double)l - 1) * (l - 1) - (double)m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
      }
    }
  }
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(rn::g_ide_c), c, sizeof(c));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(rn::g_ide_a), a, sizeof(a));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(rn::g_ide_b), b, sizeof(b));
  g_tab_status = (e == hipSuccess) ? 0 : REFNERF_EHIP;
}
int ensure_tables() {
  std::call_once(g_tab_once, upload_tables);
  return g_tab_status ? fail(REFNERF_EHIP, "uploading IDE tables failed%s") : 0;
}
}  // namespace

extern "C" {

int refnerf_abi_version(void) { return REFNERF_ABI_VERSION; }
const char *refnerf_last_error(void) { return g_err; }

void refnerf_level_cfg_default(refnerf_level_cfg *c) {
  memset(c, 0, sizeof(*c));
  c->n_samples = 128; c->n_in = 1; c->training = 0; c->compute_extras = 1;
  c->srgb_mapping = 1; c->srgb_mapping_normalization = 1; c->render_srgb_mode = REFNERF_SRGB_NONE;
  c->opaque_background = 0; c->ray_shape = 0; c->precision = REFNERF_PREC_F32;
  c->anneal = 1.0f; c->resample_padding = 0.01f; c->s_near = 0.0f; c->s_far = 1.0f;
  c->density_bias = 0.5f; c->roughness_bias = -1.0f;
  c->rgb_premultiplier = 1.0f; c->rgb_bias = 0.0f; c->rgb_padding = 0.001f; c->bg_rgb = 1.0f;
}

int refnerf_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(REFNERF_ENOGPU, "no HIP device%s");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) return fail(REFNERF_ENOGPU, "device is %s, this library is built for gfx950 only", p.gcnArchName);
  return REFNERF_OK;
}

size_t refnerf_packed_weights_bytes(int precision) {
  if (precision == REFNERF_PREC_F32) return (size_t)rn::PACKED.total * sizeof(float);
  return 0;
}

int refnerf_pack_weights(const float *d_params, void *d_packed, int precision, void *stream) {
  if (!d_params || !d_packed) return fail(REFNERF_EINVAL, "refnerf_pack_weights: null pointer%s");
  if (precision != REFNERF_PREC_F32) return fail(REFNERF_EUNSUPPORTED, "refnerf_pack_weights: precision mode not built%s");
  dim3 grid(64, rn::NUM_OPS);
  hipLaunchKernelGGL(rn::pack_weights_f32, grid, dim3(256), 0, (hipStream_t)stream, d_params, (float *)d_packed);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

static int rays_per_wg(int N) {
  if (N % rn::T_TILE == 0) return 1;
  if (rn::T_TILE % N == 0) return rn::T_TILE / N;
  if ((2 * N) % rn::T_TILE == 0) return 2;
  if ((4 * N) % rn::T_TILE == 0) return 4;
  return 1;
}

int refnerf_level_forward(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays,
                          int32_t R, const float *d_sdist_in, const float *d_weights_in,
                          const refnerf_level_out *out, void *stream) {
  if (!d_packed || !cfg || !rays || !out || !d_sdist_in || !d_weights_in)
    return fail(REFNERF_EINVAL, "refnerf_level_forward: null pointer%s");
  if (R <= 0) return fail(REFNERF_EINVAL, "refnerf_level_forward: R must be positive%s");
  /* stepfun.py:234-235 */
  if (cfg->n_samples <= 1) return fail(REFNERF_EINVAL, "num_samples must be > 1%s");
  /* render.py:126 */
  if (cfg->ray_shape != 0 && cfg->ray_shape != 1) return fail(REFNERF_EINVAL, "ray_shape must be 'cone' or 'cylinder'%s");
  if (cfg->n_in < 1 || cfg->n_in > 512) return fail(REFNERF_EINVAL, "n_in must be in [1,512]%s");
  if (cfg->precision != REFNERF_PREC_F32) return fail(REFNERF_EUNSUPPORTED, "precision mode not built%s");
  if (cfg->training) return fail(REFNERF_EUNSUPPORTED, "training-mode level (density-gradient normals) not built yet%s");
  if (!rays->d_origins || !rays->d_directions || !rays->d_viewdirs || !rays->d_radii || !rays->d_near || !rays->d_far)
    return fail(REFNERF_EINVAL, "refnerf_level_forward: null ray field%s");
  int rc = ensure_tables();
  if (rc) return rc;
  const int N = cfg->n_samples;
  const int rpw = rays_per_wg(N);
  if (rpw * N > 640) return fail(REFNERF_EINVAL, "n_samples too large for the LDS budget (rays_per_wg*N must be <= 640)%s");
  size_t lds = sizeof(float) * (size_t)(rn::DIR_PAD * rn::T_TILE + rn::HD_ROWS * rn::T_TILE + 2 * rpw * (N + 1) + rn::NPS * rpw * N + 8);
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute((const void *)rn::level_fwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  rn::LevelArgs a;
  a.packed = (const float *)d_packed;
  a.cfg = *cfg;
  a.rays = *rays;
  a.R = R;
  a.rpw = rpw;
  a.sdist_in = d_sdist_in;
  a.weights_in = d_weights_in;
  a.out = *out;
  int grid = (R + rpw - 1) / rpw;
  hipStream_t st = (hipStream_t)stream;
  const bool timed = g_timing && g_events_used < 65536;
  if (timed) {
    if (g_events_used == g_events.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      g_events.emplace_back(e0, e1);
    }
    HIP_TRY(hipEventRecord(g_events[g_events_used].first, st));
  }
  hipLaunchKernelGGL(rn::level_fwd_f32, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  HIP_TRY(hipGetLastError());
  if (timed) {
    HIP_TRY(hipEventRecord(g_events[g_events_used].second, st));
    g_events_used += 1;
  }
  return REFNERF_OK;
}

int refnerf_sample_intervals(const float *d_t, const float *d_logits, int32_t R, int32_t M, int32_t N,
                             float s_min, float s_max, float *d_sdist, int32_t *d_bin_idx, void *stream) {
  if (!d_t || !d_logits || !d_sdist) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: null pointer%s");
  if (N <= 1) return fail(REFNERF_EINVAL, "num_samples must be > 1%s");
  if (M < 1 || M > 2048 || N > 2048 || R <= 0) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: size out of range%s");
  size_t lds = sizeof(float) * 4 * (size_t)((M + 1) + M + (M + 1) + N + (N + 1) + 8);
  if (lds > 160 * 1024) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: M,N too large%s");
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute((const void *)rn::sample_intervals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL(rn::sample_intervals_kernel, dim3((R + 3) / 4), dim3(256), lds, (hipStream_t)stream,
                     d_t, d_logits, R, M, N, s_min, s_max, d_sdist, d_bin_idx);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_integrated_pos_enc(const float *d_lmean, const float *d_lvar, int32_t n, float *d_feat, void *stream) {
  if (!d_lmean || !d_lvar || !d_feat || n <= 0) return fail(REFNERF_EINVAL, "refnerf_integrated_pos_enc: bad argument%s");
  hipLaunchKernelGGL(rn::ipe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_lmean, d_lvar, n, d_feat);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_integrated_dir_enc(const float *d_xyz, const float *d_kappa_inv, int32_t n, float *d_ide, void *stream) {
  if (!d_xyz || !d_kappa_inv || !d_ide || n <= 0) return fail(REFNERF_EINVAL, "refnerf_integrated_dir_enc: bad argument%s");
  int rc = ensure_tables();
  if (rc) return rc;
  hipLaunchKernelGGL(rn::ide_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_xyz, d_kappa_inv, n, d_ide);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_set_timing(int enable) {
  g_timing = enable != 0;
  g_events_used = 0;
  return REFNERF_OK;
}
int refnerf_get_timing(double *total_ms, int64_t *launches) {
  double tot = 0.0;
  for (size_t i = 0; i < g_events_used; ++i) {
    HIP_TRY(hipEventSynchronize(g_events[i].second));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g_events[i].first, g_events[i].second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)g_events_used;
  return REFNERF_OK;
}

}  /* extern "C" */
