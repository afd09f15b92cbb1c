/*
 * refnerf_sq_train.hip -- second translation unit of librefnerf_hip.so: the round-5 training path of the parity-grade 16-bit
 * mode (REFNERF_PREC_F16X2, built-in IPE basis) on the eval kernel's skeleton.  Weight image, ACT / DELTA formats:
 * refnerf_sq_layout.h; kernels: refnerf_level_sq_fwd.h, refnerf_level_sq_bwd.h, refnerf_wgrad_sq.h.
 * Entry points (C++, internal): refnerf_sq_host.h; the C ABI stays in refnerf_hip.hip.
 */
#define REFNERF_SECONDARY_TU 1
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "refnerf_hip.h"
#include "refnerf_sq_host.h"
#include "refnerf_level_common.h"
#include "refnerf_level_bf16.h"
#include "refnerf_sq_layout.h"
#include "refnerf_level_sq_fwd.h"
#include "refnerf_level_sq_bwd.h"
#include "refnerf_wgrad_sq.h"
#include "refnerf_pack_common.h"

namespace rn {

/* ------------------------------------------------------------------ */
/* weight image of the training kernels                               */
/* ------------------------------------------------------------------ */

/* One transposed chunk of the 16x16x32 sections (the VJP): A[row = input feature col0 + 32 ob + 16 T + r][k] = W[k][row] of
 * forward op `op`, k in the k-step order of the forward's spatial chunks (refnerf_layout.h), no bias */
__device__ inline void fill_chunk_sqt(const float *__restrict__ P, char *__restrict__ chunk, int op, int col0, int ob, int kind) {
  for (int e = threadIdx.x; e < 256; e += blockDim.x) reinterpret_cast<float *>(chunk)[e] = 0.0f;
  for (int idx = threadIdx.x; idx < 16 * 512; idx += blockDim.x) {
    const int e = idx & 7, lane = (idx >> 3) & 63, pi = idx >> 9;
    const int bk = lane >> 4, r16 = lane & 15;
    const int sl = pi >> 2, which = pi & 3;
    const int T = which & 1, part = which >> 1;
    const int st = (kind == SQ_B ? 4 : 0) + sl;
    const int k = 32 * st + 16 * (e >> 2) + 4 * bk + (e & 3);
    const float v = canon_w(P, op, k, col0 + ob * 32 + 16 * T + r16);
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16 *>(chunk + 1024)[idx] = part ? (_Float16)(v - (float)hi) : hi;
  }
}
/* One plain chunk of the directional trunk of the training forward: fill_chunk_plain with every weight replaced by its hi
 * (part 0) or lo (part 1) half; the bias piece only in the hi chunk that opens the slice */
__device__ inline void fill_chunk_plain_part(const float *__restrict__ P, char *__restrict__ chunk, int op, int ob, int kind, bool first, int base, int part) {
  for (int e = threadIdx.x; e < 256; e += blockDim.x) {
    float v = 0.0f;
    if (e < 32 && first) {
      int reg = e & 15, h = e >> 4;
      v = canon_b(P, op, ob * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
    }
    reinterpret_cast<float *>(chunk)[e] = v;
  }
  for (int idx = threadIdx.x; idx < 16 * 512; idx += blockDim.x) {
    int e = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9;
    int h = lane >> 5, row = ob * 32 + (lane & 31);
    float v = 0.0f;
    const bool reg_step = (kind == BF_REG) || (kind == BF_BNLDS && t < 8);
    if (reg_step) {
      int r = 8 * (t & 1) + e;
      int feat = 32 * (t >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
      v = canon_w(P, op, row, (kind == BF_REG) ? feat : base + feat);
    } else {
      int kp = 16 * (t - 8) + 8 * h + e;
      /* dir k': [Re x36 | n.v | 0 0 0 | Im x36 | 0 0 0 0] */
      if (kp < IDE_TERMS) v = canon_w(P, op, row, base + BNECK + kp);
      else if (kp == IDE_TERMS) v = canon_w(P, op, row, base + BNECK + IDE_DIM);
      else if (kp >= 40 && kp < 40 + IDE_TERMS) v = canon_w(P, op, row, base + BNECK + IDE_TERMS + (kp - 40));
    }
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16 *>(chunk + 1024)[idx] = part ? (_Float16)(v - (float)hi) : hi;
  }
}
/* One plain TRANSPOSED chunk (backward): A[row = col0 + 32 ob + lane % 32][k] = W[k][row] of forward op `op`;
 * REG: k = the 256 outputs of the layer in register-step order; HEADS (kind BF_BNLDS): k = head rows, 0..127 in register-step
 * order, then 128 + k' (k' = 16 (t - 8) + 8 h + e < 11) from the LDS tile */
__device__ inline void fill_chunk_plain_t(const float *__restrict__ P, char *__restrict__ chunk, int op, int col0, int ob, int kind, int part) {
  for (int e = threadIdx.x; e < 256; e += blockDim.x) reinterpret_cast<float *>(chunk)[e] = 0.0f;
  for (int idx = threadIdx.x; idx < 16 * 512; idx += blockDim.x) {
    int e = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9;
    int h = lane >> 5, row = col0 + ob * 32 + (lane & 31);
    float v = 0.0f;
    if (kind == BF_REG || t < 8) {
      int r = 8 * (t & 1) + e;
      int k = 32 * (t >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
      v = canon_w(P, op, k, row);
    } else {
      int kp = 16 * (t - 8) + 8 * h + e;
      if (kp < HROWS - BNECK) v = canon_w(P, op, BNECK + kp, row);
    }
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16 *>(chunk + 1024)[idx] = part ? (_Float16)(v - (float)hi) : hi;
  }
}

/* grid = TR_CHUNKS blocks of 256 threads: block c fills chunk c (refnerf_sq_layout.h) */
__global__ void pack_train_chunks(const float *__restrict__ P, char *__restrict__ out) {
  const int c = blockIdx.x;
  char *chunk = out + (size_t)c * BF_CHUNK_BYTES;
  if (c < TR_SP_TRUNK) {
    /* spatial ops 0..7 as in the eval image */
    int op, ob, j;
    if (c < 8) { op = 0; ob = c; j = 0; }
    else if (c < 72) { op = 1 + (c - 8) / 16; ob = ((c - 8) % 16) / 2; j = (c - 8) & 1; }
    else if (c < 96) { op = 5; ob = (c - 72) / 3; j = (c - 72) % 3; }
    else { op = 6 + (c - 96) / 16; ob = ((c - 96) % 16) / 2; j = (c - 96) & 1; }
    const int kind = (op == 0 || j == 2) ? SQ_X : (j == 0 ? SQ_A : SQ_B);
    fill_chunk_sq(P, chunk, op, ob, kind, j == 0, op == 5 ? WIDTH : 0);
    return;
  }
  if (c < TR_SP_TRUNK + TR_HEADS) {
    const int i = c - TR_SP_TRUNK;
    if (i < 8) fill_chunk_sq(P, chunk, OP_HEADS, i >> 1, (i & 1) ? SQ_B : SQ_A, (i & 1) == 0, 0);
    else fill_chunk_sq(P, chunk, OP_HEADS, 4, SQ_SC, true, 0);
    return;
  }
  if (c < TR_RUN) {
    /* VJP: L7 L6 [L5 ipe] L5 L4 L3 L2 L1 [L0 ipe] */
    int i = c - TR_SP_TRUNK - TR_HEADS;
    int op, col0;
    if (i < 32) { op = 7 - i / 16; col0 = 0; i %= 16; }
    else if (i < 38) { op = 5; col0 = WIDTH; i -= 32; }
    else if (i < 118) { op = 5 - (i - 38) / 16; col0 = 0; i = (i - 38) % 16; }
    else { op = 0; col0 = 0; i -= 118; }
    fill_chunk_sqt(P, chunk, op, col0, i >> 1, (i & 1) ? SQ_B : SQ_A);
    return;
  }
  if (c < TR_FWD) {
    /* directional trunk: vd0 8 x [BN hi lo] | vd1..4 8 x [REG hi lo] | vd5 8 x [REG hi lo BN hi lo] | vd6, 7 | rgb [REG hi lo] */
    int i = c - TR_RUN;
    int op, ob, kind, part;
    bool first;
    if (i < 16) { op = 9; ob = i >> 1; kind = BF_BNLDS; part = i & 1; first = part == 0; }
    else if (i < 80) { op = 10 + (i - 16) / 16; ob = ((i - 16) % 16) >> 1; kind = BF_REG; part = i & 1; first = part == 0; }
    else if (i < 112) { op = 14; ob = (i - 80) >> 2; const int j = (i - 80) & 3; kind = j < 2 ? BF_REG : BF_BNLDS; part = j & 1; first = j == 0; }
    else if (i < 144) { op = 15 + (i - 112) / 16; ob = ((i - 112) % 16) >> 1; kind = BF_REG; part = i & 1; first = part == 0; }
    else { op = OP_RGB; ob = 0; kind = BF_REG; part = i & 1; first = part == 0; }
    fill_chunk_plain_part(P, chunk, op, ob, kind, first, op == 14 ? WIDTH : 0, part);
    return;
  }
  {
    /* backward: dir layers 7..1 | the 204 dir-input rows: 7 x [layer 5 hi lo | layer 0 hi lo] | heads^T 8 x [hi lo] | spatial 7..1 */
    int i = c - TR_BWD0;
    if (i < TR_BWD_DIR) { fill_chunk_plain_t(P, chunk, 9 + 7 - i / 16, 0, (i % 16) >> 1, BF_REG, i & 1); return; }
    i -= TR_BWD_DIR;
    if (i < TR_BWD_DIN) { const int j = i & 3; fill_chunk_plain_t(P, chunk, j < 2 ? 14 : 9, j < 2 ? WIDTH : 0, i >> 2, BF_REG, j & 1); return; }
    i -= TR_BWD_DIN;
    if (i < TR_BWD_HEADS) { fill_chunk_plain_t(P, chunk, OP_HEADS, 0, i >> 1, BF_BNLDS, i & 1); return; }
    i -= TR_BWD_HEADS;
    fill_chunk_plain_t(P, chunk, 7 - i / 16, 0, (i % 16) >> 1, BF_REG, i & 1);
  }
}

/* the constants block: W_density, W_rgb, and per transposed op G = max over its output rows f of sum_o |W[o][f]|;
 * grid = TRG_N blocks of 1024 threads: thread (f, rg) sums rows o = rg (mod 4) of column f with four loads in flight
 * (one thread per column walking 256 rows one dependent load at a time took 88 us of every training step; now 27) */
__global__ __launch_bounds__(1024) void pack_train_consts(const float *__restrict__ P, float *__restrict__ K) {
  __shared__ float red[1024];
  const int g = blockIdx.x, f = threadIdx.x & 255, rg = threadIdx.x >> 8;
  int op = -1, rows = WIDTH, col0 = 0, ncol = WIDTH;
  if (g >= TRG_SP + 1 && g <= TRG_SP + 7) op = g - TRG_SP;
  else if (g == TRG_SP5_IPE) { op = 5; col0 = WIDTH; ncol = IPE_DIM; }
  else if (g == TRG_SP0) { op = 0; ncol = IPE_DIM; }
  else if (g >= TRG_VD + 1 && g <= TRG_VD + 7) op = 9 + g - TRG_VD;
  else if (g == TRG_VD5_DIN) { op = 14; col0 = WIDTH; ncol = DIR_IN; }
  else if (g == TRG_VD0) { op = 9; ncol = DIR_IN; }
  else if (g == TRG_HEADS) { op = OP_HEADS; rows = HROWS; }
  else if (g == TRG_RGB) { op = OP_RGB; rows = 3; }
  else if (g == TRG_WD) { op = OP_HEADS; rows = 1; }
  float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (op >= 0 && f < ncol) {
    for (int o0 = rg; o0 < rows; o0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + 4 * u;
        if (o < rows) s[u] += fabsf(canon_w(P, op, g == TRG_WD ? HROW_DENSITY : o, col0 + f));
      }
    }
  }
  red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (rg == 0) red[f] = (red[f] + red[f + 256]) + (red[f + 512] + red[f + 768]);
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) K[TRC_G + g] = red[0];
  if (g == 0 && rg == 0) {
    K[TRC_WD + f] = canon_w(P, OP_HEADS, HROW_DENSITY, f);
    for (int c = 0; c < 3; ++c) K[TRC_WRGB + c * WIDTH + f] = canon_w(P, OP_RGB, c, f);
    if (f < TRC_WRGB - TRC_G - TRG_N) K[TRC_G + TRG_N + f] = 0.0f;
  }
}

}  // namespace rn

/* ================================================================== */
/* host                                                               */
/* ================================================================== */

#define SQ_HIP_TRY(expr)                                                     \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) return rnh::fail(REFNERF_EHIP, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

namespace rnsq {

size_t image_bytes() { return rn::TR_IMAGE_BYTES; }

int pack(const float *d_params, void *d_packed, hipStream_t st) {
  hipLaunchKernelGGL(rn::pack_train_chunks, dim3(rn::TR_CHUNKS), dim3(256), 0, st, d_params, (char *)d_packed);
  hipLaunchKernelGGL(rn::pack_train_consts, dim3(rn::TRG_N), dim3(1024), 0, st, d_params,
                     (float *)((char *)d_packed + rn::TR_CONST_OFF));
  SQ_HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

static int rays_per_wg_sq(int N) {
  if (N % rn::BT == 0) return 1;
  if (rn::BT % N == 0) return rn::BT / N;
  if ((2 * N) % rn::BT == 0 && 2 * N <= 512) return 2;
  if ((4 * N) % rn::BT == 0 && 4 * N <= 512) return 4;
  return 1;
}
static size_t fwd_lds_bytes(int rays, int N) {
  const size_t per_wg = sizeof(float) * (size_t)(2 * rays * (N + 1) + rn::NPS_TRAIN * rays * N + 3 * rn::BT + 8 + 12 * rays);
  return (size_t)rn::BF_RING_BYTES + rn::BF_X_BYTES + sizeof(float) * rn::HD_ROWS * rn::BT + per_wg;
}

int forward(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays, int R, const float *d_sdist_in,
            const float *d_weights_in, const refnerf_level_out *out, float *d_act, hipStream_t st) {
  const int N = cfg->n_samples;
  int rpw = rays_per_wg_sq(N);
  /* the per-ray phases (resample, compositing) occupy one wave per ray: as many rays per workgroup as the LDS holds */
  while (2 * rpw <= rn::BF_NW && 2 * rpw * N <= 640 && fwd_lds_bytes(2 * rpw, N) + (size_t)rnh::lds_pad() <= 160 * 1024 && R / (2 * rpw) >= 512) rpw *= 2;
  while (rpw > 1 && fwd_lds_bytes(rpw, N) + (size_t)rnh::lds_pad() > 160 * 1024) rpw /= 2;
  size_t lds = fwd_lds_bytes(rpw, N) + (size_t)rnh::lds_pad();
  if (lds > 160 * 1024) return rnh::fail(REFNERF_EINVAL, "n_samples too large for the 160 KiB LDS budget of the REFNERF_PREC_F16X2 training forward%s");
  {
    const int nw = rpw < rn::BF_NW ? rpw : rn::BF_NW;
    const size_t scratch = sizeof(float) * (size_t)nw * (3 * (cfg->n_in + 4) + N + 3);
    if (scratch > (size_t)rn::BF_X_BYTES) return rnh::fail(REFNERF_EINVAL, "n_in / n_samples too large for the resampler scratch of this precision mode%s");
  }
  static hipError_t attr = [] {
    hipError_t e = hipFuncSetAttribute((const void *)rn::level_fwd_train_sq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e != hipSuccess ? e : hipFuncSetAttribute((const void *)rn::level_fwd_train_sq_h, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }();
  if (attr != hipSuccess) return rnh::fail(REFNERF_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr));
  rn::LevelArgs a;
  a.packed = d_packed;
  a.cfg = *cfg;
  a.rays = *rays;
  a.R = R;
  a.rpw = rpw;
  a.sdist_in = d_sdist_in;
  a.weights_in = d_weights_in;
  a.out = *out;
  a.prof = nullptr;
  a.g_means = nullptr; a.g_covs = nullptr; a.cov_full = 0;
  a.act = d_act; a.act_pitch = 0;
  a.ring_off = 0;
  if (rnh::prof_on()) {
    int prc = rnh::prof_buffer(&a.prof);
    if (prc) return prc;
  }
  const int grid = (R + rpw - 1) / rpw;
  long tslot = -1;
  { int trc = rnh::timer_begin(st, &tslot, REFNERF_TIMER_FORWARD); if (trc) return trc; }
  if (cfg->wgrad_mode == REFNERF_WGRAD_F16) hipLaunchKernelGGL(rn::level_fwd_train_sq_h, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else hipLaunchKernelGGL(rn::level_fwd_train_sq, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  SQ_HIP_TRY(hipGetLastError());
  { int trc = rnh::timer_end(st, tslot); if (trc) return trc; }
  if (a.prof) {
    long long hbuf[8 * 32];
    SQ_HIP_TRY(hipMemcpy(hbuf, a.prof, sizeof(hbuf), hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; ++w) {
      fprintf(stderr, "[prof sq-fwd] wave %d:", w);
      for (int sl = 1; sl <= 18; ++sl) fprintf(stderr, " %lld", hbuf[w * 32 + sl] ? hbuf[w * 32 + sl] - hbuf[w * 32] : -1LL);
      fprintf(stderr, "\n");
    }
  }
  return REFNERF_OK;
}

int backward_chain(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays, int R, const float *d_sdist,
                   const refnerf_level_grads *grads, const float *d_act, float *d_delta, const float *d_seeds, long long pitch,
                   hipStream_t st) {
  (void)d_sdist;
  static hipError_t attr = hipFuncSetAttribute((const void *)rn::level_bwd_sq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return rnh::fail(REFNERF_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr));
  rn::SqBwdArgs a;
  a.packed = d_packed;
  a.cfg = *cfg;
  a.viewdirs = rays->d_viewdirs;
  a.S = (long long)R * cfg->n_samples;
  /* passes per workgroup: at least two workgroups per CU in flight over the launch, at most 8 passes (the prologue is short) */
  long long total = (a.S + rn::BT - 1) / rn::BT;
  int passes = (int)(total / 1024);
  passes = passes < 1 ? 1 : (passes > 8 ? 8 : passes);
  a.passes = passes;
  a.g_s_diffuse = grads->d_g_diffuse; a.g_s_specular = grads->d_g_specular; a.g_s_tint = grads->d_g_tint; a.g_s_rough = grads->d_g_roughness;
  a.act = d_act;
  a.delta = d_delta;
  a.seeds = d_seeds;
  a.pitch = pitch;
  a.prof = nullptr;
  if (rnh::prof_on()) {
    int prc = rnh::prof_buffer(&a.prof);
    if (prc) return prc;
    SQ_HIP_TRY(hipMemset(a.prof, 0, 8 * 32 * sizeof(long long)));
  }
  const int grid = (int)((total + passes - 1) / passes);
  const size_t lds = (size_t)rn::SQB_LDS_BYTES + (size_t)rnh::lds_pad();
  long tslot = -1;
  { int trc = rnh::timer_begin(st, &tslot, REFNERF_TIMER_BACKWARD); if (trc) return trc; }
  hipLaunchKernelGGL(rn::level_bwd_sq, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  SQ_HIP_TRY(hipGetLastError());
  { int trc = rnh::timer_end(st, tslot); if (trc) return trc; }
  if (a.prof) {
    long long hbuf[8 * 32];
    SQ_HIP_TRY(hipMemcpy(hbuf, a.prof, sizeof(hbuf), hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; ++w) {
      fprintf(stderr, "[prof sq-bwd] wave %d:", w);
      for (int sl = 1; sl <= 9; ++sl) fprintf(stderr, " %lld", hbuf[w * 32 + sl] ? hbuf[w * 32 + sl] - hbuf[w * 32] : -1LL);
      fprintf(stderr, "  | counted-wait %lld barrier-wait %lld\n", hbuf[w * 32 + 20], hbuf[w * 32 + 21]);
    }
  }
  return REFNERF_OK;
}

int wgrad(const float *d_act, const float *d_delta, long long S, long long pitch, int k_per_slice, int slices, float *d_part,
          float *d_kmin, int act11, int *slices_used, hipStream_t st) {
  (void)pitch;
  *slices_used = slices;
  static hipError_t attr = hipFuncSetAttribute((const void *)rn::wgrad_sq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rn::SQW_LDS);
  if (attr != hipSuccess) return rnh::fail(REFNERF_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr));
  static hipError_t attr2 = hipFuncSetAttribute((const void *)rn::wgrad_sq256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rn::SQ2_LDS);
  if (attr2 != hipSuccess) return rnh::fail(REFNERF_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr2));
  /* REFNERF_WGRAD_SQ_TILE=128: the four-wave 128 x 128 tiles (measurement aid; the default 256 x 256 tile fetches every operand once) */
  static const bool tile128 = [] { const char *e = getenv("REFNERF_WGRAD_SQ_TILE"); return e && atoi(e) == 128; }();
  SQ_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_kmin, 0x7f800000, 32, st));
  hipLaunchKernelGGL(rn::delta_kappa_min, dim3(256), dim3(256), 0, st, d_delta, S, d_kmin);
  rn::WgradSqArgs w;
  w.act = d_act; w.delta = d_delta; w.S = S; w.k_per_slice = k_per_slice; w.part = d_part;
  long tslot = -1;
  { int trc = rnh::timer_begin(st, &tslot, REFNERF_TIMER_WGRAD); if (trc) return trc; }
  if (tile128) {
    const dim3 grid(8 * ((slices + 7) / 8) * rn::WJOBS_SQ.tiles);
    hipLaunchKernelGGL(rn::wgrad_sq_kernel, grid, dim3(64 * rn::SQW_NW), rn::SQW_LDS, st, w, slices, d_kmin, act11);
  } else {
    /* One workgroup per CU (160 KB of LDS), so jobs x slices workgroups run in rounds of #CUs and a partial last round leaves
     * most of the chip idle behind compute-bound stragglers: the slice count is cut to whole rounds (20 jobs x 32 slices =
     * 2.5 rounds on 256 CUs -> 25 slices = 500 workgroups: 2.52 -> 2.10 ms at 4096 x 128; 24: 2.23, 28: 2.85, 19: 2.80).
     * REFNERF_WGRAD_SQ_SLICES overrides (measurement aid). */
    static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    static const int want = [] { const char *e = getenv("REFNERF_WGRAD_SQ_SLICES"); return e ? atoi(e) : 0; }();
    int se = slices;
    if (rn::WJOBS_SQ.n * se > cus) se = (rn::WJOBS_SQ.n * se / cus) * cus / rn::WJOBS_SQ.n;
    if (want > 0 && want <= slices) se = want;
    const long long blocks = (S + rn::RB - 1) / rn::RB;
    const long long per = (blocks + se - 1) / se;
    se = (int)((blocks + per - 1) / per);
    w.k_per_slice = (int)(per * rn::RB);
    *slices_used = se;
    hipLaunchKernelGGL(rn::wgrad_sq256_kernel, dim3(rn::WJOBS_SQ.n * se), dim3(512), rn::SQ2_LDS, st, w, se, d_kmin, act11);
  }
  SQ_HIP_TRY(hipGetLastError());
  { int trc = rnh::timer_end(st, tslot); if (trc) return trc; }
  return REFNERF_OK;
}

}  // namespace rnsq
