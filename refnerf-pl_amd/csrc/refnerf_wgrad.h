/*
 * refnerf_wgrad.h -- weight/bias gradients of one level from the backward
 * workspace: dW[o][k] = sum_s DELTA[o][s] * ACT[k][s], db[o] = sum_s DELTA[o][s]
 * (the contraction PyTorch autograd performs for every nn.Linear of
 * internal/models.py:497-531).
 *
 * This file: the job table shared by both arithmetic modes, the fp32-MFMA kernel
 * (cfg.wgrad_mode = REFNERF_WGRAD_F32) and the fixed-order slice reduction; the
 * default split-bf16 kernel is refnerf_wgrad_bf16x3.h.
 *
 * One fp32-MFMA GEMM over all 18 layers: a constexpr job table maps (DELTA row
 * range, ACT row range) to the canonical gradient blob; a workgroup owns one
 * 128x128 output tile and one slice of the sample axis (split-K), accumulates
 * it on v_mfma_f32_32x32x2_f32 from LDS-staged 128x32 operand tiles, and
 * writes its partial tile to PART[slice].  wgrad_reduce then adds the slices
 * in fixed order into the gradient blob: bit-reproducible, no atomics.
 */
#pragma once
#include "refnerf_level_common.h"

namespace rn {

struct WJob { int d_row, n_out, a_row, n_in, w_off, ld, b_off, tiles_n, tile0; };
constexpr int WG_TM = 128, WG_TN = 128, WG_KT = 64, WG_LDK = 68;
constexpr int MAX_WJOBS = 32;
struct WJobs { WJob job[MAX_WJOBS]; int n; int tiles; };

/* TM: rows of DELTA (output rows) per tile -- 128 for the fp32 / split-bf16 GEMMs and the 4-wave f16 GEMM, 256 for the 8-wave one */
template <int TM>
constexpr WJobs make_wjobs_t() {
  WJobs J{};
  int n = 0, t = 0;
  auto add = [&](int d_row, int n_out, int a_row, int n_in, int w_off, int ld, int b_off) {
    WJob j{};
    j.d_row = d_row; j.n_out = n_out; j.a_row = a_row; j.n_in = n_in; j.w_off = w_off; j.ld = ld; j.b_off = b_off;
    j.tiles_n = (n_in + WG_TN - 1) / WG_TN;
    j.tile0 = t;
    t += ((n_out + TM - 1) / TM) * j.tiles_n;
    J.job[n++] = j;
  };
  for (int i = 0; i < DEPTH; ++i) {            /* spatial MLP (models.py:576-580) */
    const int dr = DEL_SP + i * WIDTH, ld = CANON.sp_in[i];
    if (i == 0) add(dr, WIDTH, ACT_IPE, IPE_DIM, CANON.sp_w[0], ld, CANON.sp_b[0]);
    else {
      add(dr, WIDTH, ACT_SP + (i - 1) * WIDTH, WIDTH, CANON.sp_w[i], ld, CANON.sp_b[i]);
      if (i == 5) add(dr, WIDTH, ACT_IPE, IPE_DIM, CANON.sp_w[i] + WIDTH, ld, -1);
    }
  }
  /* heads (models.py:582,613,634-645): ONE job over the 139 contiguous head rows of DELTA, so that the
   * shared input x7 is read once instead of six times; the six tensors sit at different places of the
   * canonical blob, hence a row -> offset table (HEAD_ROWS, marked by ld = 0) instead of (w_off, ld) */
  add(DEL_HEADS, HROWS, ACT_SP + 7 * WIDTH, WIDTH, 0, 0, 0);
  for (int i = 0; i < DEPTH; ++i) {            /* directional MLP (models.py:690-694) */
    const int dr = DEL_VD + i * WIDTH, ld = CANON.vd_in[i];
    if (i == 0) add(dr, WIDTH, ACT_DIN, DIR_IN, CANON.vd_w[0], ld, CANON.vd_b[0]);
    else {
      add(dr, WIDTH, ACT_VD + (i - 1) * WIDTH, WIDTH, CANON.vd_w[i], ld, CANON.vd_b[i]);
      if (i == 5) add(dr, WIDTH, ACT_DIN, DIR_IN, CANON.vd_w[i] + WIDTH, ld, -1);
    }
  }
  add(DEL_RGB, 3, ACT_VD + 7 * WIDTH, WIDTH, CANON.rgb_w, WIDTH, CANON.rgb_b);   /* models.py:699 */
  J.n = n; J.tiles = t;
  return J;
}
constexpr WJobs WJOBS = make_wjobs_t<WG_TM>();
constexpr WJobs WJOBS_M256 = make_wjobs_t<256>();
/* general IPE basis: the tail W_ext[L][256][EXT_K] (refnerf_layout.h) = deltas of layer 0 / 5 x the tail matrix's 576 rows;
 * offsets relative to the tail (its partials and its part of the gradient blob are addressed from NUM_PARAMS on) */
constexpr WJobs make_wjobs_ext() {
  WJobs J{};
  int t = 0;
  for (int L = 0; L < 2; ++L) {
    WJob j{};
    j.d_row = DEL_SP + (L ? 5 : 0) * WIDTH; j.n_out = WIDTH; j.a_row = 0; j.n_in = EXT_K; j.w_off = L * WIDTH * EXT_K; j.ld = EXT_K; j.b_off = -1;
    j.tiles_n = (EXT_K + WG_TN - 1) / WG_TN;
    j.tile0 = t;
    t += ((WIDTH + WG_TM - 1) / WG_TM) * j.tiles_n;
    J.job[L] = j;
  }
  J.n = 2; J.tiles = t;
  return J;
}
constexpr WJobs WJOBS_EXT = make_wjobs_ext();

struct HeadRows { int w[HROWS], b[HROWS]; };
constexpr HeadRows make_head_rows() {
  HeadRows H{};
  auto put = [&](int row0, int n, int w_off, int b_off) {
    for (int i = 0; i < n; ++i) { H.w[row0 + i] = w_off + i * WIDTH; H.b[row0 + i] = b_off + i; }
  };
  put(0, BNECK, CANON.bneck_w, CANON.bneck_b);
  put(HROW_DENSITY, 1, CANON.density_w, CANON.density_b);
  put(HROW_GRAD, 3, CANON.gradpred_w, CANON.gradpred_b);
  put(HROW_ROUGH, 1, CANON.rough_w, CANON.rough_b);
  put(HROW_DIFFUSE, 3, CANON.diffuse_w, CANON.diffuse_b);
  put(HROW_TINT, 3, CANON.tint_w, CANON.tint_b);
  return H;
}
constexpr HeadRows HEAD_ROWS = make_head_rows();
/* blob offsets of output row `orow` of job J: weights row start, bias element */
__device__ __forceinline__ size_t wjob_row_off(const WJob &J, int orow) {
  return J.ld ? (size_t)J.w_off + (size_t)orow * J.ld : (size_t)HEAD_ROWS.w[orow];
}
__device__ __forceinline__ int wjob_bias_off(const WJob &J, int orow) { return J.ld ? J.b_off + orow : HEAD_ROWS.b[orow]; }

struct WgradArgs {
  const float *act, *delta;   /* blocked rows (refnerf_layout.h: [64-sample block][unit][64]) */
  int a_units, d_units;       /* unit counts of the two matrices in their formats = block strides / 64 */
  long long pitch;
  long long S;            /* valid samples (columns) */
  int k_per_slice;        /* samples per split-K slice, multiple of WG_KT */
  float *part;            /* [slices][NUM_PARAMS] */
};

#ifndef REFNERF_SECONDARY_TU   /* (kernels of the first translation unit) */
/* grid = (WJOBS.tiles, slices), 256 threads: waves 2x2 over the 128x128 tile. */
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs A) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float *Ds = wsm, *As = wsm + WG_TM * WG_LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  int ji = 0;
#pragma unroll 1
  for (int j = 1; j < WJOBS.n; ++j) if ((int)blockIdx.x >= WJOBS.job[j].tile0) ji = j;
  const WJob J = WJOBS.job[ji];
  const int tl = blockIdx.x - J.tile0;
  const int tm = tl / J.tiles_n, tn = tl - tm * J.tiles_n;
  const long long k_begin = (long long)blockIdx.y * A.k_per_slice;
  long long k_end = k_begin + A.k_per_slice;
  const long long s_pad = (A.S + WG_KT - 1) / WG_KT * WG_KT;   /* <= pitch; columns >= S hold zeros */
  if (k_end > s_pad) k_end = s_pad;

  v16f acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;    /* loader: rows lrow + 16p (p = 0..7), 4 samples at lc4 */
  /* per-thread row pointers (NULL = outside the job: zero rows) */
  const float *dp[8], *ap[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int orow = tm * WG_TM + lrow + 16 * p, irow = tn * WG_TN + lrow + 16 * p;
    dp[p] = (orow < J.n_out) ? A.delta + (long long)(J.d_row + orow) * RB + lc4 : nullptr;
    ap[p] = (irow < J.n_in) ? A.act + (long long)(J.a_row + irow) * RB + lc4 : nullptr;
  }
  static_assert(WG_KT == RB, "one k-step = one 64-sample block of the operand matrices");
  const long long dstep = (long long)A.d_units, astep = (long long)A.a_units;   /* floats per sample of k0: block k0/64 starts at k0 * units */
  v4f dv[8], av[8];
  /* plain loads, nothing consumes them before the next rendezvous (a tail mask on the loaded values
   * would put an s_waitcnt right behind every load): the columns [S, pitch) of both matrices are
   * zeroed by wgrad_zero_tail, and slices end on multiples of WG_KT */
  auto fetch = [&](long long k0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      v4f x = {0.0f, 0.0f, 0.0f, 0.0f}, y = {0.0f, 0.0f, 0.0f, 0.0f};
      if (dp[p]) x = *reinterpret_cast<const v4f *>(dp[p] + k0 * dstep);
      if (ap[p]) y = *reinterpret_cast<const v4f *>(ap[p] + k0 * astep);
      dv[p] = x; av[p] = y;
    }
  };
  if (k_begin < k_end) fetch(k_begin);
  for (long long k0 = k_begin; k0 < k_end; k0 += WG_KT) {
    __syncthreads();                                   /* previous tile fully consumed */
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      *reinterpret_cast<v4f *>(Ds + (lrow + 16 * p) * WG_LDK + lc4) = dv[p];
      *reinterpret_cast<v4f *>(As + (lrow + 16 * p) * WG_LDK + lc4) = av[p];
      bsum[p] += (dv[p][0] + dv[p][1]) + (dv[p][2] + dv[p][3]);
    }
    __syncthreads();
    if (k0 + WG_KT < k_end) fetch(k0 + WG_KT);         /* next tile's loads fly under this tile's MFMAs */
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < WG_KT / 8; ++kk) {
      v4f a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const v4f *>(Ds + (wm * 64 + i * 32 + sl) * WG_LDK + kk * 8 + 4 * h);
        b[i] = *reinterpret_cast<const v4f *>(As + (wn * 64 + i * 32 + sl) * WG_LDK + kk * 8 + 4 * h);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float *part = A.part + (size_t)blockIdx.y * NUM_PARAMS;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int colk = tn * WG_TN + wn * 64 + j * 32 + sl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = tm * WG_TM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < J.n_out && colk < J.n_in) part[wjob_row_off(J, orow) + colk] = acc[i][j][r];
      }
    }
  if (tn == 0 && J.b_off >= 0) {
    /* bias gradient: the 16 loader threads of a row hold its partial sums */
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float s = bsum[p];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      const int orow = tm * WG_TM + lrow + 16 * p;
      if ((tid & 15) == 0 && orow < J.n_out) part[wjob_bias_off(J, orow)] = s;
    }
  }
}

/* zero the pad samples [S, pitch) of the first `rows` units of a blocked matrix of `units` units */
__global__ void wgrad_zero_tail(float *m, int rows, int units, long long pitch, long long S) {
  const int tail = (int)(pitch - S);
  const long long n = (long long)rows * tail;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m[(i / tail) * RB + rb_col(S + (i % tail), units)] = 0.0f;
}

/* zero units [u0, u1) of a blocked matrix of `units` units over every sample column [0, pitch): the rows of the tail matrix
 * that belong to direction groups a basis of fewer than 7 groups does not have (the training forward never writes them) */
__global__ void wgrad_zero_units(float *m, int u0, int u1, int units, long long pitch) {
  const long long n = (long long)(u1 - u0) * pitch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m[(u0 + i / pitch) * RB + rb_col(i % pitch, units)] = 0.0f;
}

/* grads[i] += sum over slices (fixed order) of PART[slice][i] */
/* (two parameters per thread as one 8-byte load per slice -- n and every slice offset are even -- and four slices' loads in
 * flight: 111 MB in 76 -> 27 us; the order of the additions per parameter is unchanged: slice 0, 1, 2, ...) */
__global__ void wgrad_reduce(const float *__restrict__ part, int slices, float *__restrict__ grads, int n) {
  typedef float rv2 __attribute__((ext_vector_type(2)));
  const int n2 = n >> 1;
  if (((n & 1) == 0) && ((reinterpret_cast<size_t>(part) | reinterpret_cast<size_t>(grads)) & 7) == 0) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) {
      const rv2 *p = reinterpret_cast<const rv2 *>(part) + i;
      rv2 s = {0.0f, 0.0f};
      int c = 0;
      for (; c + 4 <= slices; c += 4) {
        const rv2 a0 = p[(size_t)c * n2], a1 = p[(size_t)(c + 1) * n2], a2 = p[(size_t)(c + 2) * n2], a3 = p[(size_t)(c + 3) * n2];
        s = s + a0; s = s + a1; s = s + a2; s = s + a3;
      }
      for (; c < slices; ++c) s = s + p[(size_t)c * n2];
      rv2 *gp = reinterpret_cast<rv2 *>(grads) + i;
      *gp = *gp + s;
    }
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int c = 0; c < slices; ++c) s += part[(size_t)c * n + i];
    grads[i] += s;
  }
}

#endif

}  // namespace rn
