/*
 * refnerf_level_bwd_f32.h -- backward of one level (fp32 MFMA).
 *
 * Same decomposition as the forward kernel (workgroup = 4 waves = RPW whole
 * rays, 32-sample blocks per wave, transposed GEMMs with activations in the
 * accumulator registers).  The training forward saved every linear layer's
 * input (the ACT matrix); here only the head and rgb rows are recomputed from
 * the saved x7 / v7, every ReLU mask comes from the bit masks the training
 * forward saved behind ACT (4 dwords per lane and layer, all 8 layers of a
 * trunk loaded up front), and the gradient runs
 *   rendering -> alpha weights -> density / sample rgb / predicted normals
 *   -> colour head -> directional MLP -> IDE / reflection -> heads -> spatial MLP
 * through the transposed packed weights (refnerf_layout.h: TOP_*).
 *
 * Weight gradients are NOT accumulated here: every layer's pre-activation
 * gradient (DELTA) is streamed to HBM as a [feature][sample] matrix like ACT
 * -- one row per k-step through the store hook of the GEMM that consumes it
 * (RowStoreHook, non-temporal) -- and refnerf_wgrad.h / refnerf_wgrad_bf16x3.h
 * contract the two over the sample axis.
 *
 * Restates the autograd of internal/models.py:533-750 + render.py:132-216
 * (SURVEY.md A10); oracle: rn_level_train / rn_level_backward.
 */
#pragma once
#include "refnerf_level_f32.h"
#include "refnerf_level_bf16.h"

namespace rn {

struct BwdArgs {
  const void *packed;
  refnerf_level_cfg cfg;
  refnerf_rays rays;
  int R;
  int rpw;
  const float *sdist;       /* [R,N+1] this level's sample edges (forward output) */
  const float *density;     /* [R,N]   forward history                            */
  const float *rgb;         /* [R,N,3]                                            */
  const float *weights;     /* [R,N]                                              */
  const float *g_r_rgb;     /* [R,3]   dL/d rendering rgb (after the render map)  */
  const float *g_weights;   /* [R,N]   dL/d history weights, or NULL              */
  const float *g_npred;     /* [R,N,3] dL/d history normals_pred, or NULL         */
  const float *g_r_acc;     /* [R]     dL/d rendering acc, or NULL                */
  const float *g_r_dist;    /* [R]     dL/d rendering distance, or NULL           */
  /* optional per-sample seeds on the other differentiable history outputs (models.py:731-750) */
  const float *g_s_density; /* [R,N]   dL/d history density                        */
  const float *g_s_rgb;     /* [R,N,3] dL/d history rgb (padded colour)            */
  const float *g_s_diffuse; /* [R,N,3] dL/d history diffuse                        */
  const float *g_s_specular;/* [R,N,3] dL/d history specular                       */
  const float *g_s_tint;    /* [R,N,3] dL/d history tint                           */
  const float *g_s_rough;   /* [R,N]   dL/d history roughness                      */
  const float *act;         /* [ACT_ROWS][pitch] layer inputs saved by the training forward */
  float *delta;             /* [DEL_ROWS][pitch] written here                       */
  float *seeds;             /* [NGS][pitch] per-sample seeds: bwd_seed_kernel -> level_bwd_* */
  long long pitch;
  int ring_off;             /* byte offset of the shared weight-stream ring in dynamic LDS (bf16 chains) */
  int act16;                /* ACT holds bf16 rows (written by the bf16-chain training forward), masks stay 32-bit */
  long long *prof;          /* debug: per-phase cycle stamps of workgroup 0 (REFNERF_PROF=1), else NULL */
};

/* A-fragment prefetch depth of the chain GEMMs: the stores riding in the same in-order vmcnt queue make the
 * effective latency of an A load longer than in the forward */
#ifndef REFNERF_PF_BWD
#define REFNERF_PF_BWD 3
#endif
constexpr int PF_BWD = REFNERF_PF_BWD;
/* DELTA as bf16 rows in the bf16-chain backward (the chain deltas are bf16-exact; the head / rgb rows get rounded) */
#ifndef REFNERF_DELTA16
#define REFNERF_DELTA16 1
#endif
constexpr int NGS = 7;      /* per-sample upstream gradients (SEEDS rows): density, rgb[3], n_pred[3] */

/* Per-ray part of the backward (one wave per ray): rendering gradient through
 * the render-time colour map (render.py:186-216), compositing (152-176) and
 * the alpha weights (132-149) down to per-sample gradients in GS. */
__device__ __forceinline__ void bwd_seed_rays(const BwdArgs &A, float *TD, int ray0, int rpw, int wave, int lane) {
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  for (int rl = wave; rl < rpw; rl += 4) {
    const int ray = ray0 + rl;
    if (ray >= A.R) break;
    float *td = TD + rl * (N + 1);
    const float nearv = A.rays.d_near[ray], farv = A.rays.d_far[ray];
    for (int k = lane; k <= N; k += 64) td[k] = s_to_t(A.sdist[(size_t)ray * (N + 1) + k], nearv, farv, A.cfg.raydist);
    wave_sync();
    const float dx = A.rays.d_directions[(size_t)ray * 3], dy = A.rays.d_directions[(size_t)ray * 3 + 1],
                dz = A.rays.d_directions[(size_t)ray * 3 + 2];
    const float norm_d = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float *wg = A.weights + (size_t)ray * N, *dg = A.density + (size_t)ray * N, *cg = A.rgb + (size_t)ray * N * 3;
    const int C = (N + 63) / 64, i0 = lane * C;
    float acc = 0.0f, s_rgb[3] = {0.0f, 0.0f, 0.0f};
    for (int i = i0; i < i0 + C && i < N; ++i) {
      const float w = wg[i];
      acc += w;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] += w * cg[i * 3 + c];
    }
    acc = wave_sum(acc);
    float pre[3], g_rgb[3];
    const float bg_w = fmaxf(0.0f, 1.0f - acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) { pre[c] = wave_sum(s_rgb[c]) + bg_w * cfg.bg_rgb; g_rgb[c] = A.g_r_rgb[(size_t)ray * 3 + c]; }
    const int mode = cfg.render_srgb_mode;
    if (mode != REFNERF_SRGB_NONE)
      colour_map_backward(pre, mode == REFNERF_SRGB_NORM_LINEAR || mode == REFNERF_SRGB_NORM_SRGB,
                          mode == REFNERF_SRGB_SRGB || mode == REFNERF_SRGB_NORM_SRGB, g_rgb);
    const float gsum = (g_rgb[0] + g_rgb[1] + g_rgb[2]) * cfg.bg_rgb;
    /* acc = sum_i w_i and distance = sum_i w_i t_mid,i are linear in the weights (render.py:161-176) */
    const float g_acc = A.g_r_acc ? A.g_r_acc[ray] : 0.0f, g_dist = A.g_r_dist ? A.g_r_dist[ray] : 0.0f;
    auto dLdw = [&](int i) {
      float g = (g_rgb[0] * cg[i * 3] + g_rgb[1] * cg[i * 3 + 1]) + g_rgb[2] * cg[i * 3 + 2];
      if (acc < 1.0f) g -= gsum;                              /* bg weight = max(0, 1 - acc) */
      if (A.g_weights) g += A.g_weights[(size_t)ray * N + i];
      g += g_acc + g_dist * (0.5f * (td[i] + td[i + 1]));
      return g;
    };
    /* dL/dw_i and the two running sums of the weight backward */
    double l_dd = 0.0, l_gw = 0.0;
    for (int i = i0; i < i0 + C && i < N; ++i) {
      const float g = dLdw(i);
      l_dd += (double)(dg[i] * ((td[i + 1] - td[i]) * norm_d));
      l_gw += (double)(g * wg[i]);
    }
    const double incl_dd = wave_scan_incl(l_dd, lane), incl_gw = wave_scan_incl(l_gw, lane);
    const double tot_gw = __shfl(incl_gw, 63, 64);
    double cum = incl_dd - l_dd;                              /* optical depth in front of sample i */
    double run_gw = incl_gw - l_gw;
    for (int i = i0; i < i0 + C && i < N; ++i) {
      const float w = wg[i];
      const float g = dLdw(i);
      const float delta = (td[i + 1] - td[i]) * norm_d;
      const float dd = dg[i] * delta;
      run_gw += (double)(g * w);
      const double suffix = tot_gw - run_gw;                  /* sum_{j>i} g_j w_j */
      /* w_i = (1 - e^{-dd_i}) e^{-cum_i} */
      const float g_dd = g * expf(-dd) * expf(-(float)cum) - (float)suffix;
      cum += (double)dd;
      float g_density = g_dd * delta;
      if (cfg.opaque_background && i == N - 1) g_density = 0.0f;
      float *gs = A.seeds + ((size_t)ray * N + i);
      gs[0] = g_density + (A.g_s_density ? A.g_s_density[(size_t)ray * N + i] : 0.0f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gs[(size_t)(1 + c) * A.pitch] = w * g_rgb[c] + (A.g_s_rgb ? A.g_s_rgb[((size_t)ray * N + i) * 3 + c] : 0.0f);
        gs[(size_t)(4 + c) * A.pitch] = A.g_npred ? A.g_npred[((size_t)ray * N + i) * 3 + c] : 0.0f;
      }
    }
  }
}

__device__ __forceinline__ v16f load_acc_blk(__amdgpu_buffer_rsrc_t rs, int off, int h, int ob) {
  v4f b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, h * 64 + q * 16, off * 4 + ob * 128, 0));
  return (v16f){b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], b[1][2], b[1][3],
                b[2][0], b[2][1], b[2][2], b[2][3], b[3][0], b[3][1], b[3][2], b[3][3]};
}

/* block 4 (the 12 scalar head rows) of the fp32 heads op alone: same k order and accumulation as gemm_op<5, 8, true>
 * gives that block, a fifth of the MFMAs and of the A stream (plane 1, first dword of each lane's 16 bytes) */
__device__ __forceinline__ void heads_scalar_block_f32(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h,
                                                       const v16f (&in)[8], v16f &o) {
  constexpr int STEP_BYTES = 64 * 8 * 4, D = 4;
  const int voff = lane * 16;
  const int soff = a_off * 4 + 1024;
  float a[D];
#pragma unroll
  for (int d = 0; d < D; ++d) a[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + d * STEP_BYTES, 0));
  v16f acc[1];
  load_acc<1>(rs, b_off + 4 * 32, h, acc);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int step = 0; step < REG_STEPS; ++step) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step % D], in[step >> 4][step & 15], acc[0], 0, 0, 0);
    /* (loads past the op's last k-step stay inside the image: the next op follows) */
    a[step % D] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + (step + D) * STEP_BYTES, 0));
    __builtin_amdgcn_sched_barrier(0);
  }
  o = acc[0];
}

/* the saved ReLU sign patterns of the 8 layers of one trunk (ACT_MASK rows 8*layer0 ...): 4 dwords per layer */
__device__ __forceinline__ void load_masks(const float *act, long long pitch, long long rpitch, int layer0, size_t gs, size_t acol, int h, unsigned (&M)[8][4], bool act16) {
  if (act16) {                                   /* bf16 format: one 16-B slot per layer in the sample-major block */
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const v4u w = *smb_slot(act, pitch, gs, h, SMB_MASK + layer0 + l);
#pragma unroll
      for (int q = 0; q < 4; ++q) M[l][q] = w[q];
    }
    return;
  }
  const long long e0 = (long long)(ACT_MASK + 8 * layer0 + 4 * h) * rpitch + (long long)acol;   /* one origin, compile-time row offsets */
#pragma unroll
  for (int l = 0; l < 8; ++l)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      M[l][q] = __builtin_bit_cast(unsigned, act[e0 + (long long)(8 * l + q) * rpitch]);
}


/* rolled layer loops cannot index the mask registers dynamically: the next layer's mask moves up to M[7] */
__device__ __forceinline__ void shift_masks(unsigned (&M)[8][4]) {
#pragma unroll
  for (int l = 7; l > 0; --l)
#pragma unroll
    for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
}

/* BF: transposed chains on bf16 MFMA; SP: on split-f16 operands (22-bit deltas and weights, hi*hi + lo*hi + hi*lo on
 * v_mfma_f32_32x32x16_f16: the parity-grade fast chains, see refnerf_level_f32.h), fp32 ACT / DELTA rows */
/* PF (split chains only): the split-f16 formats of refnerf_layout.h -- ACT read as hi / lo pair units (REFNERF_ACT_F16X2), DELTA
 * written as ONE half per element (the hi halves of the packed deltas as they are: delta * c_s at 11 bits) plus the factor
 * c_s per (layer id, sample) in the DSC units.  Half the DELTA bytes, and no arithmetic between the fragments and HBM. */
template <bool BF, bool SP = false, bool PF = false>
__device__ __forceinline__ void level_bwd_body(const BwdArgs &A) {
  static_assert(!(BF && SP), "one chain arithmetic");
  static_assert(!PF || SP, "pair formats belong to the split chains");
  constexpr bool D16 = BF && (REFNERF_DELTA16 != 0);
  constexpr int DUNITS = PF ? DEL_UNITS_F16S : del_units(D16);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int rpw = A.rpw;
  const int ray0 = blockIdx.x * rpw;
  const int n_tot = rpw * N;

  float *X = smem;                               /* [DIR_PAD][T_TILE] */
  float *HD = X + DIR_PAD * T_TILE;              /* [HD_ROWS][T_TILE] */
  float *TD = HD + HD_ROWS * T_TILE;             /* [rpw][N+1]        */
  char *ring = reinterpret_cast<char *>(smem) + A.ring_off;   /* bf16 chains: the shared weight-stream ring (RING_BYTES) */

  RN_STAMP(A, 0);
  for (int rl = wave; rl < rpw; rl += 4) {       /* the t-distances of this workgroup's rays */
    const int ray = ray0 + rl;
    if (ray >= A.R) break;
    const float nearv = A.rays.d_near[ray], farv = A.rays.d_far[ray];
    for (int k = lane; k <= N; k += 64) TD[rl * (N + 1) + k] = s_to_t(A.sdist[(size_t)ray * (N + 1) + k], nearv, farv, A.cfg.raydist);
  }
  __syncthreads();
  RN_STAMP(A, 1);

  const int col = wave * 32 + sl;
  const float *xl = X + h * T_TILE + col;
  v16f in[8], out[8];
  unsigned M[8][4];                              /* ReLU masks of the trunk being walked (saved by the training forward) */

  for (int pass0 = 0; pass0 < n_tot; pass0 += T_TILE) {
    /* The image descriptor is rebuilt from a laundered pointer in every pass: with a loop-invariant descriptor the
     * compiler hoisted ~100 weight / bias loads of the pass in front of the loop (one pass per workgroup at the usual
     * shapes: nothing gained) and spilled them -- 24 k cycles of spill traffic before the first pass began. */
    const void *packed_l = A.packed;
    long long pitch = A.pitch;                     /* same for the row pitch: ~140 hoisted 64-bit row origins, all spilled */
    asm volatile("" : "+s"(packed_l), "+s"(pitch));
    constexpr long long rpitch = RB;               /* unit pitch of the blocked ACT / DELTA rows (refnerf_layout.h) */
    int hdb = DIR_PAD * T_TILE + col;                /* the HD tile (beyond the 64 KB immediate range) through one laundered base */
    asm volatile("" : "+v"(hdb));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)packed_l, 0, PACKED.total * 4, 0x00020000);
    RN_STAMP(A, 14);
    const int g = pass0 + col;
    const int rl = g / N, si = g - rl * N;
    const int ray = ray0 + rl;
    const bool valid = (g < n_tot) && (ray < A.R);
    const int rayc = valid ? ray : (A.R - 1);
    const size_t gs = valid ? (size_t)ray * N + si : 0;
    /* this sample's column in the blocked rows of DELTA and (fp32 format) ACT; the sample-major block and the seed
     * rows keep (pitch, gs) */
    const size_t dcol = (size_t)rb_col((long long)gs, DUNITS), acol = (size_t)rb_col((long long)gs, ACT_UNITS_F32);
    /* PF: one packed pair (value0, value1) * c rounded to halves -> pair-row `row` / 2 of DELTA; the factor of layer `lid` */
    auto store_dpair = [&](int row, float x0, float x1, float c) {
      const v2hf pv = __builtin_convertvector((v2f){x0 * c, x1 * c}, v2hf);
      stream_store_u(A.delta, (long long)(row >> 1) * rpitch + (long long)dcol, __builtin_bit_cast(unsigned, pv));
    };
    /* (0 for a sample without any gradient in that layer: it takes no part in the layer's smallest factor) */
    auto store_dscale = [&](int lid, float c, bool live) {
      if (valid && h == 0) stream_store(A.delta + (long long)(DSC0 + lid) * rpitch + (long long)dcol, live ? c : 0.0f);
    };
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = A.rays.d_viewdirs[(size_t)rayc * 3 + i];
    /* ===== the few forward values the backward needs, from the saved x7 / v7 ===== */
    /* Only the scalar block of the heads (rows 128..139: density, grad_pred, roughness, diffuse, tint) is needed here --
     * the bottleneck enters the backward only as a weight-gradient operand (its ACT_DIN rows).  bf16 chains on a bf16 forward: the forward's own GEMM (same
     * image, same packed x7) -> the forward's own raw head values, bit for bit. */
    {
      v16f hd4[1];
      bool done = false;
      if constexpr (BF) {
        if (A.act16) {
          v4uu xpk[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) { const v4u w = *smb_slot(A.act, pitch, gs, h, SMB_X7 + t); xpk[t] = (v4uu){w[0], w[1], w[2], w[3]}; }
          gemm_op_bf16<1, 16, 0, true>(rs, PACKED.bf_off[OP_HEADS] + 4 * 256, PACKED.op[OP_HEADS].b_off + 4 * 32, lane, h, xpk, hd4, nullptr);
          done = true;
        }
      }
      if constexpr (PF) {
        /* split-f16 pair units: x7 comes back as the very fragments the forward's heads GEMM consumed -- the same GEMM on them
         * (block 4 of the split image, same product order) gives the forward's raw head values bit for bit, at 48 f16 MFMAs
         * instead of 128 fp32 ones and without widening 256 values */
        v4uu xh[16], xw[16];
        load_frags_split(A.act, rpitch, ACT_SP + 7 * WIDTH, acol, h, xh, xw);
        gemm_op_split<1, 16, 0, true>(rs, PACKED.hf_off[OP_HEADS] + 4 * 256, PACKED.op[OP_HEADS].b_off + 4 * 32, lane, h, xh, xw, hd4, nullptr);
        done = true;
      }
      if (!done) {
        if (A.act16) smb_load_rows(A.act, pitch, gs, h, SMB_X7, in);                  /* x7: input of the heads */
        else load_rows<8>(A.act, rpitch, ACT_SP + 7 * WIDTH, acol, h, in);
        heads_scalar_block_f32(rs, PACKED.op[OP_HEADS].a_off, PACKED.op[OP_HEADS].b_off, lane, h, in, hd4[0]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < HD_ROWS) X[hdb + row * T_TILE] = hd4[0][r];
      }
    }
    RN_STAMP(A, 2);
    bool rgb_bf = false;
    v16f rgbv[1];
    if constexpr (BF) {
      if (A.act16) {                                 /* the forward's own rgb GEMM on the packed v7 */
        v4uu xpk[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { const v4u w = *smb_slot(A.act, pitch, gs, h, SMB_V7 + t); xpk[t] = (v4uu){w[0], w[1], w[2], w[3]}; }
        gemm_op_bf16<1, 16, 0, true>(rs, PACKED.bf_off[OP_RGB], PACKED.op[OP_RGB].b_off, lane, h, xpk, rgbv, nullptr);
        rgb_bf = true;
      }
    }
    if constexpr (PF) {                              /* the forward's own rgb GEMM on its saved input fragments */
      v4uu xh[16], xw[16];
      load_frags_split(A.act, rpitch, ACT_VD + 7 * WIDTH, acol, h, xh, xw);
      gemm_op_split<1, 16, 0, true>(rs, PACKED.hf_off[OP_RGB], PACKED.op[OP_RGB].b_off, lane, h, xh, xw, rgbv, nullptr);
      rgb_bf = true;
    }
    if (!rgb_bf) {
      if (A.act16) smb_load_rows(A.act, pitch, gs, h, SMB_V7, in);                  /* v7: input of the rgb layer */
      else load_rows<8>(A.act, rpitch, ACT_VD + 7 * WIDTH, acol, h, in);
    }
    wave_sync();
    SampleHeads sh;
    float raw_density, raw_rough, raw_tint[3];
    {
      float gp[3], raw_dif[3];
      raw_density = X[hdb + 0 * T_TILE];
      raw_rough = X[hdb + 4 * T_TILE];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        gp[i] = X[hdb + (1 + i) * T_TILE];
        raw_dif[i] = X[hdb + (5 + i) * T_TILE];
        raw_tint[i] = X[hdb + (8 + i) * T_TILE];
      }
      sample_heads(cfg, raw_density, gp, raw_rough, raw_dif, raw_tint, v, sh);
    }
    float raw_rgb[3];
    {
      if (!rgb_bf) gemm_op<1, 1, true>(rs, PACKED.op[OP_RGB].a_off, PACKED.op[OP_RGB].b_off, lane, h, in, rgbv, xl, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(rgbv[0][i], sl, 64);
    }
    RN_STAMP(A, 3);
    load_masks(A.act, pitch, rpitch, 8, gs, acol, h, M, A.act16);                                    /* directional trunk */

    /* ================= backward ================= */
    float gsv[NGS];
#pragma unroll
    for (int i = 0; i < NGS; ++i) gsv[i] = valid ? A.seeds[(size_t)i * pitch + gs] : 0.0f;
    /* ---- colour head (models.py:699-729) ---- */
    float g_tint[3], g_raw_rgb[3], g_raw_diff[3];
    {
      float sg[3], dl[3], colr[3], g_col[3];
      const float pad_scale = (float)(1.0 + 2.0 * (double)cfg.rgb_padding);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        sg[i] = sigmoid_t(cfg.rgb_premultiplier * raw_rgb[i] + cfg.rgb_bias);
        dl[i] = sigmoid_t(sh.raw_dif[i] - LOG3_F);
        colr[i] = sh.tint[i] * sg[i] + dl[i];
        g_col[i] = gsv[1 + i] * pad_scale;
      }
      if (cfg.srgb_mapping) colour_map_backward(colr, cfg.srgb_mapping_normalization != 0, true, g_col);
      /* seeds on the history's own diffuse / specular (clip(srgb(.)) of the linear colours when
       * srgb_mapping, models.py:718-719, else the linear colours) and tint */
      float g_dl[3], g_sp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { g_dl[i] = g_col[i]; g_sp[i] = g_col[i]; }
      if (A.g_s_diffuse || A.g_s_specular) {
        float e_dl[3], e_sp[3], spl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          e_dl[i] = (valid && A.g_s_diffuse) ? A.g_s_diffuse[gs * 3 + i] : 0.0f;
          e_sp[i] = (valid && A.g_s_specular) ? A.g_s_specular[gs * 3 + i] : 0.0f;
          spl[i] = sh.tint[i] * sg[i];
        }
        if (cfg.srgb_mapping) {
          colour_map_backward(dl, false, true, e_dl);
          colour_map_backward(spl, false, true, e_sp);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { g_dl[i] += e_dl[i]; g_sp[i] += e_sp[i]; }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        g_tint[i] = g_sp[i] * sg[i] + ((valid && A.g_s_tint) ? A.g_s_tint[gs * 3 + i] : 0.0f);
        g_raw_rgb[i] = (g_sp[i] * sh.tint[i]) * sg[i] * (1.0f - sg[i]) * cfg.rgb_premultiplier;
        g_raw_diff[i] = g_dl[i] * dl[i] * (1.0f - dl[i]);
      }
    }
    if constexpr (PF) {
      /* (both half-waves hold the same three values) */
      bool lr;
      const float cr = pow2_scale_for(fmaxf(fmaxf(fabsf(g_raw_rgb[0]), fabsf(g_raw_rgb[1])), fabsf(g_raw_rgb[2])), &lr);
      if (valid && h == 0) { store_dpair(DEL_RGB, g_raw_rgb[0], g_raw_rgb[1], cr); store_dpair(DEL_RGB + 2, g_raw_rgb[2], 0.0f, cr); }
      store_dscale(17, cr, lr);
    } else if (valid && h == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) store_row1<D16>(A.delta, rpitch, DEL_RGB + i, dcol, g_raw_rgb[i]);
    }
    /* ---- seed of the directional chain: W_rgb^T g_raw_rgb through the last ReLU ---- */
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const v16f w0 = load_acc_blk(rs, PACKED.wrgb_off, h, ob), w1 = load_acc_blk(rs, PACKED.wrgb_off + 256, h, ob),
                 w2 = load_acc_blk(rs, PACKED.wrgb_off + 512, h, ob);
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = (w0[r] * g_raw_rgb[0] + w1[r] * g_raw_rgb[1]) + w2[r] * g_raw_rgb[2];
    }
    RN_STAMP(A, 4);
    v4uu pk[(BF || SP) ? 16 : 1];                  /* packed delta (bf16 chains) / its hi halves (split chains) */
    v4uu pl[SP ? 16 : 1];                          /* lo halves (split chains) */
    float cs = 1.0f;                               /* split chains: this sample's power-of-two factor on the packed delta (mask_split) */
    bool dlive = true;                             /* ... and whether the sample has any non-zero delta in the current layer */
    if constexpr (SP) mask_split(out, M[7], pk, pl, cs, &dlive);
    else if constexpr (BF) mask_pack(out, M[7], pk);
    else masked_into(out, in, M[7]);
    /* ---- directional MLP, layers 7..0 ---- */
    v16f(&gd)[DIN_BLOCKS] = reinterpret_cast<v16f(&)[DIN_BLOCKS]>(out);   /* gradient w.r.t. the 201 dir inputs */
    const int din_hi = tile_hi(col);
    auto park_din = [&](int i) {
      /* layer 5 (skip connection) parks its share in LDS; layer 0 adds it back */
      const float unscale = SP ? 1.0f / cs : 1.0f;     /* split chains: the GEMM result carries the delta's factor */
#pragma unroll
      for (int blk = 0; blk < DIN_BLOCKS; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (blk < 6 || row < DIR_PAD) {
            float *px = X + (blk < 4 ? row * T_TILE + col : (row - 128) * T_TILE + din_hi);
            if constexpr (SP) gd[blk][r] *= unscale;
            if (i == 0) gd[blk][r] += *px;
            *px = gd[blk][r];
          }
        }
    };
#pragma unroll 1
    for (int i = 7; i >= 0; --i) {
      /* delta_i leaves through the store hook of the GEMM that consumes it (one row per k-step) */
      if constexpr (BF) {
        std::conditional_t<D16, PairStoreHook, RowStoreHook> sh_(A.delta, rpitch, DEL_VD + i * WIDTH, dcol, h, valid);
        auto hook = [&](int t) {
          if constexpr (D16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sh_(4 * t + e, pk[t][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh_(8 * t + e, pk_elem(pk, t >> 1, 8 * (t & 1) + e));
          }
        };
        if (i == 5) { gemm_op_bf16<DIN_BLOCKS, 16, 0, false>(rs, PACKED.bt_off[TOP_VD5_DIN], 0, lane, h, pk, gd, nullptr); park_din(5); }
        if (i == 0) { gemm_op_bf16<DIN_BLOCKS, 16, 0, false>(rs, PACKED.bt_off[TOP_VD0], 0, lane, h, pk, gd, nullptr, hook); park_din(0); }
        if (i > 0) {
          if (i == 7) RN_STAMP(A, 8);
          gemm_chain_bf16_shared<false>(rs, PACKED.bt_off[TOP_VD1 + i - 1], 0, lane, h, wave, pk, out, ring, hook);
          if (i == 7) RN_STAMP(A, 9);
          shift_masks(M);
          mask_pack(out, M[7], pk);
          if (i == 7) RN_STAMP(A, 10);
        }
      } else if constexpr (SP) {
        std::conditional_t<PF, PairStoreHook, RowStoreHook> sh_(A.delta, rpitch, DEL_VD + i * WIDTH, dcol, h, valid);
        const float inv = 1.0f / cs;
        if constexpr (PF) store_dscale(9 + i, cs, dlive);  /* delta_i leaves as its packed hi halves, carrying cs */
        auto hook = [&](int t, int quarter = -1) {
          if constexpr (PF) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (quarter < 0 || e == quarter) sh_(4 * t + e, pk[t][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (quarter < 0 || (e >> 1) == quarter) sh_(8 * t + e, split_elem(pk[t], pl[t], e) * inv);
          }
        };
        if (i == 5) { gemm_op_split<DIN_BLOCKS, 16, 0, false>(rs, PACKED.ht_off[TOP_VD5_DIN], 0, lane, h, pk, pl, gd, nullptr); park_din(5); }
        if (i == 0) { gemm_op_split<DIN_BLOCKS, 16, 0, false>(rs, PACKED.ht_off[TOP_VD0], 0, lane, h, pk, pl, gd, nullptr, hook); park_din(0); }
        if (i > 0) {
          if constexpr (PF && REFNERF_SPLIT_SHARED != 0) gemm_chain_split_shared<false>(rs, PACKED.ht_off[TOP_VD1 + i - 1], 0, lane, h, wave, pk, pl, out, ring, hook);
          else gemm_op_split<8, 16, 0, false>(rs, PACKED.ht_off[TOP_VD1 + i - 1], 0, lane, h, pk, pl, out, nullptr, hook);
          shift_masks(M);
          mask_split(out, M[7], pk, pl, cs, &dlive);
        }
      } else {
        if (i == 5 || i == 0) {
          if (i == 0) gemm_op<DIN_BLOCKS, 8, true, false>(rs, PACKED.top[TOP_VD0].a_off, 0, lane, h, in, gd, xl, 0,
                                                          RowStoreHook(A.delta, rpitch, DEL_VD, dcol, h, valid));
          else gemm_op<DIN_BLOCKS, 8, true, false>(rs, PACKED.top[TOP_VD5_DIN].a_off, 0, lane, h, in, gd, xl, 0);
          park_din(i);
        }
        if (i > 0) {
          gemm_op<8, 8, true, false, RowStoreHook, PF_BWD>(rs, PACKED.top[TOP_VD1 + i - 1].a_off, 0, lane, h, in, out, xl, 0,
                                     RowStoreHook(A.delta, rpitch, DEL_VD + i * WIDTH, dcol, h, valid));
          shift_masks(M);                                                     /* M[7] <- mask of layer i-1 */
          masked_into(out, in, M[7]);
        }
      }
    }
    RN_STAMP(A, 5);
    /* X rows 0..127: dL/d bottleneck (= head rows 0..127), rows 128..200: dL/d (IDE, n.v) */
    if constexpr (!PF) store_rows<4, D16>(A.delta, rpitch, DEL_HEADS, dcol, h, valid, gd);     /* (PF: from the tile, with the head block's factor, below) */
    wave_sync();
    /* ---- IDE, reflection, predicted normal, head activations (models.py:611-686) ---- */
    {
      float g_ref[3], g_rough;
      const int xhi = tile_hi(col);              /* rows >= 128 of the tile through one laundered base (tile_idx) */
      auto gq = [&](int q, float &) { return X[tile_idx(BNECK + q, col, xhi)]; };
      if (cfg.dir_enc == REFNERF_DIRENC_POSENC) { posenc_grad(sh.refd[0], sh.refd[1], sh.refd[2], gq, g_ref); g_rough = 0.0f; }   /* coord.pos_enc: no roughness */
      else ide_grad(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, gq, g_ref, g_rough);
      const float g_dot = X[tile_idx(BNECK + IDE_DIM, col, xhi)];
      const float w3[3] = {-v[0], -v[1], -v[2]};
      const float ndw = (sh.npred[0] * w3[0] + sh.npred[1] * w3[1]) + sh.npred[2] * w3[2];
      const float grn = (g_ref[0] * sh.npred[0] + g_ref[1] * sh.npred[1]) + g_ref[2] * sh.npred[2];
      float g_np[3], g_gp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) g_np[i] = gsv[4 + i] + 2.0f * (grn * w3[i] + ndw * g_ref[i]) + g_dot * v[i];
      /* n_pred = -g / sqrt(max(|g|^2, eps)) (ref_utils.py:40-42) */
      const float nrm2 = (sh.gp[0] * sh.gp[0] + sh.gp[1] * sh.gp[1]) + sh.gp[2] * sh.gp[2];
      const float s = fmaxf(nrm2, EPS32), rsq = sqrtf(s);
      const float gdotg = (sh.gp[0] * g_np[0] + sh.gp[1] * g_np[1]) + sh.gp[2] * g_np[2];
      const float live = (nrm2 > EPS32) ? 1.0f : (nrm2 == EPS32 ? 0.5f : 0.0f);
#pragma unroll
      for (int i = 0; i < 3; ++i) g_gp[i] = -(g_np[i] / rsq - live * sh.gp[i] * gdotg / (s * rsq));
      if (valid && A.g_s_rough) g_rough += A.g_s_rough[gs];
      const float g_raw_rough = g_rough * softplus_grad(raw_rough + cfg.roughness_bias);
      const float g_raw_density = gsv[0] * softplus_grad(raw_density + cfg.density_bias);
      float hrow[11];
      hrow[0] = g_raw_density;
      hrow[4] = g_raw_rough;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        hrow[1 + i] = g_gp[i];
        hrow[5 + i] = g_raw_diff[i];
        hrow[8 + i] = g_tint[i] * sh.tint[i] * (1.0f - sh.tint[i]);
      }
      wave_sync();                               /* both half-waves have read the IDE gradients */
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
          X[tile_idx(HROW_DENSITY + i, col, xhi)] = hrow[i];
          if constexpr (!PF) { if (valid) store_row1<D16>(A.delta, rpitch, DEL_HEADS + HROW_DENSITY + i, dcol, hrow[i]); }
        }
      } else {
#pragma unroll
        for (int q = HROWS; q < 2 * HEADS_T_STEPS + 2; ++q) X[tile_idx(q, col, xhi)] = 0.0f;
      }
    }
    wave_sync();
    /* ---- heads^T, then the spatial MLP, layers 7..0 ---- */
    RN_STAMP(A, 6);
    load_masks(A.act, pitch, rpitch, 0, gs, acol, h, M, A.act16);                                    /* spatial trunk */
    if constexpr (BF) {
      gemm_op_bf16<8, 0, BT_HEADS_STEPS, false>(rs, PACKED.bt_off[TOP_HEADS], 0, lane, h, pk, out, X + col);
      mask_pack(out, M[7], pk);
#pragma unroll 1
      for (int i = 7; i >= 0; --i) {
        if (i > 0) {
          std::conditional_t<D16, PairStoreHook, RowStoreHook> sh_(A.delta, rpitch, DEL_SP + i * WIDTH, dcol, h, valid);
          gemm_chain_bf16_shared<false>(rs, PACKED.bt_off[i - 1], 0, lane, h, wave, pk, out, ring, [&](int t) {
            if constexpr (D16) {
#pragma unroll
              for (int e = 0; e < 4; ++e) sh_(4 * t + e, pk[t][e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) sh_(8 * t + e, pk_elem(pk, t >> 1, 8 * (t & 1) + e));
            }
          });
          shift_masks(M);
          mask_pack(out, M[7], pk);
        } else {                                                             /* no GEMM consumes delta_0 */
#pragma unroll
          for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[blk][r] = pk_elem(pk, blk, r);
          store_rows<8, D16>(A.delta, rpitch, DEL_SP, dcol, h, valid, in);
        }
      }
    } else if constexpr (SP) {
      /* the 139 head-row gradients of this sample sit in the fp32 tile: their factor first (9 k-steps x 8 rows per lane) */
      {
        float m = 0.0f;
        const int hi2 = tile_hi(col);
#pragma unroll
        for (int t = 0; t < BT_HEADS_STEPS; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const int row = 16 * t + 8 * h + e; m = fmaxf(m, fabsf(X[tile_idx(row, col, hi2)])); }
        cs = pow2_scale_for(m, &dlive);
        if constexpr (PF) {
          /* the 139 head rows (+ their zero pad row) leave from the tile with this factor: half h takes pairs h, h + 2, ... */
          if (valid) {
#pragma unroll 1
            for (int j = h; j < (HROWS + 1) / 2; j += 2)
              store_dpair(DEL_HEADS + 2 * j, X[tile_idx(2 * j, col, hi2)], (2 * j + 1 < HROWS) ? X[tile_idx(2 * j + 1, col, hi2)] : 0.0f, cs);
          }
          store_dscale(8, cs, dlive);
        }
      }
      gemm_op_split<8, 0, BT_HEADS_STEPS, false>(rs, PACKED.ht_off[TOP_HEADS], 0, lane, h, pk, pl, out, X + col, NoStepHook(), cs);
      mask_split(out, M[7], pk, pl, cs, &dlive);
#pragma unroll 1
      for (int i = 7; i >= 0; --i) {
        if constexpr (PF) store_dscale(i, cs, dlive);
        if (i > 0) {
          std::conditional_t<PF, PairStoreHook, RowStoreHook> sh_(A.delta, rpitch, DEL_SP + i * WIDTH, dcol, h, valid);
          const float inv = 1.0f / cs;
          auto hook = [&](int t, int quarter = -1) {
            if constexpr (PF) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (quarter < 0 || e == quarter) sh_(4 * t + e, pk[t][e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (quarter < 0 || (e >> 1) == quarter) sh_(8 * t + e, split_elem(pk[t], pl[t], e) * inv);
            }
          };
          if constexpr (PF && REFNERF_SPLIT_SHARED != 0) gemm_chain_split_shared<false>(rs, PACKED.ht_off[i - 1], 0, lane, h, wave, pk, pl, out, ring, hook);
          else gemm_op_split<8, 16, 0, false>(rs, PACKED.ht_off[i - 1], 0, lane, h, pk, pl, out, nullptr, hook);
          shift_masks(M);
          mask_split(out, M[7], pk, pl, cs, &dlive);
        } else if constexpr (PF) {                                           /* no GEMM consumes delta_0: its hi halves as they are */
          PairStoreHook sh_(A.delta, rpitch, DEL_SP, dcol, h, valid);
#pragma unroll
          for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) sh_(4 * t + e, pk[t][e]);
        } else {                                                             /* no GEMM consumes delta_0 */
#pragma unroll
          for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[blk][r] = split_elem(pk[2 * blk + (r >> 3)], pl[2 * blk + (r >> 3)], r & 7) * (1.0f / cs);
          store_rows<8, false>(A.delta, rpitch, DEL_SP, dcol, h, valid, in);
        }
      }
    } else {
      gemm_op<8, 8, false, false>(rs, PACKED.top[TOP_HEADS].a_off, 0, lane, h, in, out, xl, HEADS_T_STEPS);
      masked_into(out, in, M[7]);
#pragma unroll 1
      for (int i = 7; i >= 0; --i) {
        if (i > 0) {
          gemm_op<8, 8, true, false, RowStoreHook, PF_BWD>(rs, PACKED.top[i - 1].a_off, 0, lane, h, in, out, xl, 0,
                                     RowStoreHook(A.delta, rpitch, DEL_SP + i * WIDTH, dcol, h, valid));
          shift_masks(M);                                                     /* M[7] <- mask of layer i-1 */
          masked_into(out, in, M[7]);
        } else store_rows<8>(A.delta, rpitch, DEL_SP, dcol, h, valid, in);      /* no GEMM consumes delta_0 */
      }
    }
    RN_STAMP(A, 7);
    wave_sync();
  }
}

/* The per-ray part of the backward (compositing, colour map, weight backward: a dependent scan along each ray) as its own
 * launch, one wave per ray: inside the level kernel it ran on one wave of a 1-workgroup-per-CU kernel with nothing to hide
 * its load latencies behind (88 k of the 570 k cycles of a bf16 pass).  grid = ceil(R / 4), dynamic LDS 4 (N + 1) floats. */
__global__ __launch_bounds__(NTHREADS) void bwd_seed_kernel(const BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_seed_rays(A, smem, blockIdx.x * 4, 4, threadIdx.x >> 6, threadIdx.x & 63);
}

__global__ __launch_bounds__(NTHREADS) void level_bwd_f32(const BwdArgs A) { level_bwd_body<false>(A); }
/* bf16 chains (cfg.precision = REFNERF_PREC_BF16): gradients at bf16 accuracy */
__global__ __launch_bounds__(NTHREADS) void level_bwd_bf16c(const BwdArgs A) { level_bwd_body<true>(A); }
/* transposed chains on split-f16 operands (cfg.precision = REFNERF_PREC_F16X2 in refnerf_level_backward) */
__global__ __launch_bounds__(NTHREADS) void level_bwd_f16x2c(const BwdArgs A) { level_bwd_body<false, true, true>(A); }
/* the same chains on fp32 ACT / DELTA rows (REFNERF_ACT_F32 activations: an f32 training forward, or a general IPE basis) */
__global__ __launch_bounds__(NTHREADS) void level_bwd_f16x2c_r32(const BwdArgs A) { level_bwd_body<false, true>(A); }

}  // namespace rn
