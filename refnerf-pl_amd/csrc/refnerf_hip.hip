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
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "refnerf_hip.h"
#include "refnerf_level_common.h"
#include "refnerf_level_f32.h"
#include "refnerf_level_bf16.h"
#include "refnerf_level_bwd_f32.h"
#include "refnerf_wgrad.h"
#include "refnerf_wgrad_bf16x3.h"
#include "refnerf_wgrad_f16.h"
#include "refnerf_rays.h"
#include "refnerf_pack_common.h"
#include "refnerf_sq_host.h"
#include "refnerf_sq_layout.h"

namespace rn {

/* ------------------------------------------------------------------ */
/* weight packing                                                     */
/* ------------------------------------------------------------------ */


__global__ void pack_weights_f32(const float *__restrict__ P, float *__restrict__ out) {
  int op = blockIdx.y;
  if (op < NUM_OPS) {
    const Op o = PACKED.op[op];
    int n_a = (o.reg_steps + o.lds_steps) * 64 * o.stride;
    int n_b = o.nob * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_a + n_b; e += gridDim.x * blockDim.x) {
      if (e < n_a) {
        int ob, lane, step = e / (o.stride * 64);
        if (o.stride == 8) { const int rem = e % 512; ob = (rem >> 8) * 4 + (rem & 3); lane = (rem & 255) >> 2; }   /* [step][q][lane][4] */
        else { ob = e % o.stride; lane = (e / o.stride) % 64; }
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
  } else if (op < NUM_OPS + NUM_TOPS) {
    /* transposed ops: A[step][lane][ob] = W[o = k slot][in_row] (see refnerf_layout.h) */
    const int t = op - NUM_OPS;
    const Op o = PACKED.top[t];
    const TopSrc src = PACKED.top_src[t];
    int n_a = (o.reg_steps + o.lds_steps) * 64 * o.stride;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_a; e += gridDim.x * blockDim.x) {
      int ob, lane, step = e / (o.stride * 64);
        if (o.stride == 8) { const int rem = e % 512; ob = (rem >> 8) * 4 + (rem & 3); lane = (rem & 255) >> 2; }   /* [step][q][lane][4] */
        else { ob = e % o.stride; lane = (e / o.stride) % 64; }
      int h = lane >> 5, in_row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (t == TOP_HEADS) {
          int hr = 2 * step + h;                               /* head row feeding this k slot */
          v = (hr < HROWS) ? canon_w(P, OP_HEADS, hr, in_row) : 0.0f;
        } else {
          int kb = step >> 4, r = step & 15;
          int oo = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;   /* output feature feeding this k slot */
          v = canon_w(P, src.fwd_op, oo, src.col0 + in_row);
        }
      }
      out[o.a_off + e] = v;
    }
  } else if (op < NUM_OPS + 2 * NUM_TOPS) {
    /* bf16 transposed ops of the bf16-chain backward (refnerf_layout.h: bt_off) */
    const int t = op - NUM_OPS - NUM_TOPS;
    const Op o = PACKED.top[t];
    const TopSrc src = PACKED.top_src[t];
    const int steps = (t == TOP_HEADS) ? BT_HEADS_STEPS : BT_CHAIN_STEPS;
    __bf16 *dst = reinterpret_cast<__bf16 *>(out + PACKED.bt_off[t]);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < steps * 64 * 8 * 8; idx += gridDim.x * blockDim.x) {
      const int e = idx & 7, lane = (idx >> 3) & 63, ob = (idx >> 9) & 7, st = idx >> 12;   /* [step][ob][lane][8] */
      const int h = lane >> 5, in_row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (t == TOP_HEADS) {
          const int hr = 16 * st + 8 * h + e;
          v = (hr < HROWS) ? canon_w(P, OP_HEADS, hr, in_row) : 0.0f;
        } else {
          const int r = 8 * (st & 1) + e;
          const int oo = 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int n_rows = (o.nob == 8) ? WIDTH : (o.nob == 3 ? IPE_DIM : DIR_IN);   /* rows beyond are padding */
          v = (in_row < n_rows) ? canon_w(P, src.fwd_op, oo, src.col0 + in_row) : 0.0f;
        }
      }
      dst[idx] = (__bf16)v;
    }
  } else if (op < 2 * NUM_OPS + 2 * NUM_TOPS) {
    /* bf16 forward ops of the bf16-chain training forward (refnerf_layout.h: bf_off) */
    const int fo = op - NUM_OPS - 2 * NUM_TOPS;
    const Op o = PACKED.op[fo];
    const int rst = bf_reg_steps(fo), steps = rst + bf_lds_steps(fo);
    const int valid_k = (fo == 0 || fo == 5) ? IPE_DIM : DIR_IN;
    __bf16 *dst = reinterpret_cast<__bf16 *>(out + PACKED.bf_off[fo]);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < steps * 64 * 8 * 8; idx += gridDim.x * blockDim.x) {
      const int e = idx & 7, lane = (idx >> 3) & 63, ob = (idx >> 9) & 7, st = idx >> 12;   /* [step][ob][lane][8] */
      const int h = lane >> 5, row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (st < rst) {
          const int r = 8 * (st & 1) + e;
          v = canon_w(P, fo, row, 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h);
        } else {
          const int kl = 16 * (st - rst) + 8 * h + e;
          v = (kl < valid_k) ? canon_w(P, fo, row, (rst ? WIDTH : 0) + kl) : 0.0f;
        }
      }
      dst[idx] = (__bf16)v;
    }
  } else if (op < 2 * NUM_OPS + 3 * NUM_TOPS) {
    /* split-f16 transposed ops (refnerf_layout.h: ht_off): the bt_off values as hi + lo halves, [k-step][hi | lo][ob][lane][8] */
    const int t = op - 2 * NUM_OPS - 2 * NUM_TOPS;
    const Op o = PACKED.top[t];
    const TopSrc src = PACKED.top_src[t];
    const int steps = (t == TOP_HEADS) ? BT_HEADS_STEPS : BT_CHAIN_STEPS;
    _Float16 *dst = reinterpret_cast<_Float16 *>(out + PACKED.ht_off[t]);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < steps * 64 * 8 * 8; idx += gridDim.x * blockDim.x) {
      const int e = idx & 7, lane = (idx >> 3) & 63, ob = (idx >> 9) & 7, st = idx >> 12;
      const int h = lane >> 5, in_row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (t == TOP_HEADS) {
          const int hr = 16 * st + 8 * h + e;
          v = (hr < HROWS) ? canon_w(P, OP_HEADS, hr, in_row) : 0.0f;
        } else {
          const int r = 8 * (st & 1) + e;
          const int oo = 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int n_rows = (o.nob == 8) ? WIDTH : (o.nob == 3 ? IPE_DIM : DIR_IN);
          v = (in_row < n_rows) ? canon_w(P, src.fwd_op, oo, src.col0 + in_row) : 0.0f;
        }
      }
      const _Float16 hi = (_Float16)v;
      const size_t base = (size_t)st * (2 * 64 * 8 * 8) + (idx & 4095);
      dst[base] = hi;
      dst[base + 4096] = (_Float16)(v - (float)hi);
    }
  } else if (op < 3 * NUM_OPS + 3 * NUM_TOPS) {
    /* split-f16 forward ops (refnerf_layout.h: hf_off) */
    const int fo = op - 2 * NUM_OPS - 3 * NUM_TOPS;
    const Op o = PACKED.op[fo];
    const int rst = bf_reg_steps(fo), steps = rst + bf_lds_steps(fo);
    const int valid_k = (fo == 0 || fo == 5) ? IPE_DIM : DIR_IN;
    _Float16 *dst = reinterpret_cast<_Float16 *>(out + PACKED.hf_off[fo]);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < steps * 64 * 8 * 8; idx += gridDim.x * blockDim.x) {
      const int e = idx & 7, lane = (idx >> 3) & 63, ob = (idx >> 9) & 7, st = idx >> 12;
      const int h = lane >> 5, row = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (ob < o.nob) {
        if (st < rst) {
          const int r = 8 * (st & 1) + e;
          v = canon_w(P, fo, row, 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h);
        } else {
          const int kl = 16 * (st - rst) + 8 * h + e;
          v = (kl < valid_k) ? canon_w(P, fo, row, (rst ? WIDTH : 0) + kl) : 0.0f;
        }
      }
      const _Float16 hi = (_Float16)v;
      const size_t base = (size_t)st * (2 * 64 * 8 * 8) + (idx & 4095);
      dst[base] = hi;
      dst[base + 4096] = (_Float16)(v - (float)hi);
    }
  } else {
    /* WD / WRGB: raw_density.weight and rgb_layer.weight rows in accumulator layout [ob][h][16] */
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < 4 * 8 * 32; b += gridDim.x * blockDim.x) {
      int reg = b & 15, h = (b >> 4) & 1, ob = (b >> 5) & 7, which = b >> 8;
      int k = ob * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (which == 0) out[PACKED.wd_off + (b & 255)] = P[CANON.density_w + k];
      else out[PACKED.wrgb_off + (which - 1) * 256 + (b & 255)] = P[CANON.rgb_w + (which - 1) * WIDTH + k];
    }
  }
}

/* Tail of the fp32 image for a general IPE basis (refnerf_layout.h): the basis directions, the forward group ops
 * (op 0's LDS-step format: A[step][q][lane][4] = W[32 ob + lane % 32][k = 2 step + h]) and their transposes (TOP_SP0's
 * format: A[step][lane][4] = W[o(step, h)][in_row = 32 ob + lane % 32], ob < 3) of the tail weights. */
__global__ void pack_weights_ext(const float *__restrict__ P, const float *__restrict__ basis, int groups, float *__restrict__ out) {
  const int e0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const int which = blockIdx.y;              /* 0: basis, then 12 blocks each: fp32 forward, fp32 transposed, split forward, split transposed */
  if (which == 0) {
    for (int e = e0; e < 64; e += stride) out[PEXT_BASIS + e] = (e < 9 * groups) ? basis[e] : 0.0f;
    return;
  }
  const int idx = (which - 1) % (2 * EXT_GROUPS), L = idx / EXT_GROUPS, g = idx % EXT_GROUPS + 1;
  const float *W = P + ext_w_off(L, g);      /* [256 rows, pitch EXT_K][96] */
  const bool live = g < groups;
  const int kind = (which - 1) / (2 * EXT_GROUPS);   /* 0: fp32 forward, 1: fp32 transposed, 2: split forward, 3: split transposed */
  if (kind == 0) {
    float *dst = out + pext_fwd_off(L, g);
    for (int e = e0; e < PEXT_FWD_FLOATS; e += stride) {
      const int step = e >> 9, rem = e & 511, ob = (rem >> 8) * 4 + (rem & 3), lane = (rem & 255) >> 2;
      const int h = lane >> 5, row = ob * 32 + (lane & 31), k = 2 * step + h;
      dst[e] = live ? W[row * EXT_K + k] : 0.0f;
    }
  } else if (kind == 1) {
    float *dst = out + pext_t_off(L, g);
    for (int e = e0; e < PEXT_T_FLOATS; e += stride) {
      const int ob = e & 3, lane = (e >> 2) & 63, step = e >> 8;
      const int h = lane >> 5, in_row = ob * 32 + (lane & 31), kb = step >> 4, r = step & 15;
      const int oo = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[e] = (live && ob < 3) ? W[oo * EXT_K + in_row] : 0.0f;
    }
  } else {
    /* split copies: element idx = [step][ob][lane][e] of a 4096-half plane pair (hi plane, lo plane 4096 halves behind) */
    const bool fwd = kind == 2;
    const int steps = fwd ? BF_IPE_STEPS : BT_CHAIN_STEPS;
    _Float16 *dst = reinterpret_cast<_Float16 *>(out + (fwd ? pext_hf_off(L, g) : pext_ht_off(L, g)));
    for (int i = e0; i < steps * 4096; i += stride) {
      const int e = i & 7, lane = (i >> 3) & 63, ob = (i >> 9) & 7, st = i >> 12;
      const int h = lane >> 5, r32 = ob * 32 + (lane & 31);
      float v = 0.0f;
      if (live) {
        if (fwd) {
          const int kl = 16 * st + 8 * h + e;                     /* plain order over the fp32 X tile (bf_off's LDS steps) */
          v = W[r32 * EXT_K + kl];
        } else if (ob < 3) {
          const int r = 8 * (st & 1) + e;
          const int oo = 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
          v = W[oo * EXT_K + r32];
        }
      }
      const _Float16 hi = (_Float16)v;
      const size_t base = (size_t)st * (2 * 4096) + (i & 4095);
      dst[base] = hi;
      dst[base + 4096] = (_Float16)(v - (float)hi);
    }
  }
}

/* bf16 image: one block per (op, ob) slice, uniform 17 KB chunks in execution
 * order; see refnerf_layout.h. */
template <typename E>   /* E = __bf16 (REFNERF_PREC_BF16) or _Float16 (REFNERF_PREC_F16): same image layout */
__global__ void pack_weights_16(const float *__restrict__ P, char *__restrict__ out) {
  const int op = blockIdx.y, ob = blockIdx.x;
  const BfOp o = BFPACKED.op[op];
  if (ob >= o.nob) return;
  const int base = (o.nchunk == 2) ? WIDTH : 0;     /* canonical column of the first non-register input */
  for (int j = 0; j < o.nchunk; ++j)
    fill_chunk_plain<E>(P, out + (size_t)(o.chunk0 + ob * o.nchunk + j) * BF_CHUNK_BYTES, op, ob, o.kind[j], j == 0, base);
}

/* ... and a plain BNLDS chunk whose eight bottleneck k-steps follow the merged-run order of that section */
__device__ void fill_chunk_bnlds_sq(const float *__restrict__ P, char *__restrict__ chunk, int op, int ob, bool first, int base) {
  fill_chunk_plain<_Float16>(P, chunk, op, ob, BF_BNLDS, first, base);
  __syncthreads();
  for (int idx = threadIdx.x; idx < 8 * 512; idx += blockDim.x) {
    const int e = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9;
    const int h = lane >> 5, row = ob * 32 + (lane & 31);
    const int feat = 32 * (t >> 1) + 16 * (e >> 2) + 8 * h + 4 * (t & 1) + (e & 3);
    reinterpret_cast<_Float16 *>(chunk + 1024)[idx] = (_Float16)canon_w(P, op, row, base + feat);
  }
}
__global__ void pack_weights_split(const float *__restrict__ P, char *__restrict__ out) {
  const int op = blockIdx.y, ob = blockIdx.x;
  const int nob = (op == OP_HEADS) ? 5 : (op == OP_RGB ? 1 : 8);
  if (ob >= nob) return;
  int c = SPPACKED.chunk0[op];
  for (int o2 = 0; o2 < ob; ++o2) c += sp_slice_chunks(op, o2);
  char *chunk = out + (size_t)c * BF_CHUNK_BYTES;
  if (op > OP_HEADS) {
    const BfOp o = BFPACKED.op[op];
    const int base = (o.nchunk == 2) ? WIDTH : 0;
    for (int j = 0; j < o.nchunk; ++j) {
      if (o.kind[j] == BF_BNLDS) fill_chunk_bnlds_sq(P, chunk + (size_t)j * BF_CHUNK_BYTES, op, ob, j == 0, base);
      else fill_chunk_plain<_Float16>(P, chunk + (size_t)j * BF_CHUNK_BYTES, op, ob, o.kind[j], j == 0, base);
    }
    return;
  }
  const int n = sp_slice_chunks(op, ob);
  if (op == OP_HEADS) { fill_chunk_sq(P, chunk, op, ob, ob < 4 ? SQ_BN : SQ_SC, true, 0); return; }
  for (int j = 0; j < n; ++j) {
    const int kind = (op == 0 || j == 2) ? SQ_X : (j == 0 ? SQ_A : SQ_B);
    fill_chunk_sq(P, chunk + (size_t)j * BF_CHUNK_BYTES, op, ob, kind, j == 0, op == 5 ? WIDTH : 0);
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
  wave_sync();
  sample_intervals_wave<true>(t_in, lg, cw, c, M, N, smin, smax, sd, bin_idx ? bin_idx + (size_t)ray * N : nullptr, lane);
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

/* compute_alpha_weights + volumetric_rendering (render.py:132-254) on caller-supplied per-sample values: the P7
 * device code of the level kernels (composite_phase) behind its own entry.  One wave per ray, rpw rays per workgroup. */
struct RenderArgs {
  refnerf_level_cfg cfg;
  int R, rpw;
  const float *density, *tdist, *dirs, *far, *rgb, *dif, *spc, *nrm, *npred, *rough, *tint;
  refnerf_level_out out;
};
__global__ __launch_bounds__(256) void render_rays_kernel(RenderArgs ra) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = ra.cfg.n_samples, rpw = ra.rpw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *TD = smem, *XP = TD + rpw * (N + 1), *PS = XP + rpw * (N + 1), *wscr = PS + (size_t)rpw * N * NPS_TRAIN;
  const int ray0 = blockIdx.x * rpw;
  for (int e = threadIdx.x; e < rpw * (N + 1); e += NTHREADS) {
    const int rl = e / (N + 1), ray = ray0 + rl;
    if (ray < ra.R) TD[e] = ra.tdist[(size_t)ray * (N + 1) + (e - rl * (N + 1))];
  }
  for (int e = threadIdx.x; e < rpw * N; e += NTHREADS) {
    const int rl = e / N, ray = ray0 + rl;
    if (ray >= ra.R) continue;
    const size_t g = (size_t)ray * N + (e - rl * N);
    float *ps = PS + (size_t)e * NPS_TRAIN;
    ps[PS_DENSITY] = ra.density[g];
    ps[PS_ROUGH] = ra.rough ? ra.rough[g] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ps[PS_RGB + c] = ra.rgb ? ra.rgb[3 * g + c] : 0.0f;
      ps[PS_DIF + c] = ra.dif ? ra.dif[3 * g + c] : 0.0f;
      ps[PS_SPC + c] = ra.spc ? ra.spc[3 * g + c] : 0.0f;
      ps[PS_NPRED + c] = ra.npred ? ra.npred[3 * g + c] : 0.0f;
      ps[PS_TINT + c] = ra.tint ? ra.tint[3 * g + c] : 0.0f;
      ps[PS_NORMALS + c] = ra.nrm ? ra.nrm[3 * g + c] : 0.0f;
    }
  }
  __syncthreads();
  LevelArgs A{};
  A.cfg = ra.cfg;
  A.R = ra.R;
  A.rpw = rpw;
  A.rays.d_directions = ra.dirs;
  A.rays.d_far = ra.far;
  A.out = ra.out;
  composite_phase<4, false, NPS_TRAIN>(A, TD, XP, PS, rpw * N, ray0, wave, lane, wscr, nullptr);
}

/* The three Ref-NeRF losses of one level as ONE pass over the level's outputs (train_utils.py:33-88 mse data term,
 * :165-183 orientation, :186-204 predicted normals), one wave per ray:
 *   terms[ray] = { sum_c lossmult (rgb_c - gt_c)^2,  sum_i w_i min(0, n_i . (-v))^2,  sum_i w_i (1 - n_i . npred_i) }
 * with n = `normals_o` (the orientation target: normals_pred or the density normals) / `normals` (density normals,
 * detached in the reference).  The host sums over the rays and applies multipliers and normalisers. */
struct LossArgs {
  int R, N;
  const float *rgb, *gt, *lossmult, *weights, *normals_o, *normals, *normals_pred, *viewdirs;
  float *terms;                 /* forward: [R,3] */
  /* backward: upstream scalars (already multiplied by multiplier / normaliser) and the gradient tensors */
  float g_data, g_orient, g_normal;
  const float *upstream;        /* device float[3]: dL/d(data, orientation, normal term) multiplying g_*, or NULL (= 1) */
  int orient_on_pred;           /* the orientation target IS normals_pred (its gradient then reaches g_npred) */
  float *g_rgb, *g_weights, *g_npred;
};
__global__ __launch_bounds__(256) void refnerf_losses_fwd_kernel(LossArgs a) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.R) return;
  const float v0 = -a.viewdirs[ray * 3], v1 = -a.viewdirs[ray * 3 + 1], v2 = -a.viewdirs[ray * 3 + 2];
  float s_o = 0.0f, s_n = 0.0f;
  for (int i = lane; i < a.N; i += 64) {
    const size_t e = (size_t)ray * a.N + i;
    const float w = a.weights[e];
    if (a.normals_o) {
      const float ndv = (a.normals_o[3 * e] * v0 + a.normals_o[3 * e + 1] * v1) + a.normals_o[3 * e + 2] * v2;
      const float m = fminf(ndv, 0.0f);
      s_o += w * (m * m);
    }
    if (a.normals) {
      const float d = (a.normals[3 * e] * a.normals_pred[3 * e] + a.normals[3 * e + 1] * a.normals_pred[3 * e + 1]) +
                      a.normals[3 * e + 2] * a.normals_pred[3 * e + 2];
      s_n += w * (1.0f - d);
    }
  }
  s_o = wave_sum(s_o);
  s_n = wave_sum(s_n);
  if (lane == 0) {
    float d = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float r = a.rgb[ray * 3 + c] - a.gt[ray * 3 + c]; d += a.lossmult[ray] * (r * r); }
    a.terms[ray * 3] = d; a.terms[ray * 3 + 1] = s_o; a.terms[ray * 3 + 2] = s_n;
  }
}
__global__ __launch_bounds__(256) void refnerf_losses_bwd_kernel(LossArgs a) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)a.R * a.N) return;
  const int ray = (int)(e / a.N);
  const float g_data = a.g_data * (a.upstream ? a.upstream[0] : 1.0f), g_orient = a.g_orient * (a.upstream ? a.upstream[1] : 1.0f),
              g_normal = a.g_normal * (a.upstream ? a.upstream[2] : 1.0f);
  const float v0 = -a.viewdirs[ray * 3], v1 = -a.viewdirs[ray * 3 + 1], v2 = -a.viewdirs[ray * 3 + 2];
  const float w = a.weights[e];
  float gw = 0.0f, gn[3] = {0.0f, 0.0f, 0.0f};
  if (a.normals_o) {
    const float ndv = (a.normals_o[3 * e] * v0 + a.normals_o[3 * e + 1] * v1) + a.normals_o[3 * e + 2] * v2;
    const float m = fminf(ndv, 0.0f);
    gw += g_orient * (m * m);
    if (a.orient_on_pred) { const float k = g_orient * w * 2.0f * m; gn[0] += k * v0; gn[1] += k * v1; gn[2] += k * v2; }
  }
  if (a.normals) {
    const float d = (a.normals[3 * e] * a.normals_pred[3 * e] + a.normals[3 * e + 1] * a.normals_pred[3 * e + 1]) +
                    a.normals[3 * e + 2] * a.normals_pred[3 * e + 2];
    gw += g_normal * (1.0f - d);
    const float k = -g_normal * w;
#pragma unroll
    for (int c = 0; c < 3; ++c) gn[c] += k * a.normals[3 * e + c];
  }
  a.g_weights[e] = gw;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.g_npred[3 * e + c] = gn[c];
  if (e < (size_t)a.R * 3) {      /* the first 3R threads also write the rendering gradient: d/d rgb of the mse term */
    const int r = (int)(e / 3);
    a.g_rgb[e] = g_data * a.lossmult[r] * 2.0f * (a.rgb[e] - a.gt[e]);
  }
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

/* The library's only process-wide state (everything else lives in caller-owned buffers):
 *  - the opt-in kernel timer of refnerf_set_timing / refnerf_get_timing: event pairs recorded on the launch stream and
 *    resolved lazily in refnerf_get_timing(), so the timed region is not serialised by event syncs; guarded by `mu`, so
 *    level calls from several host threads stay safe while it is on;
 *  - two debug knobs read from the environment ONCE (REFNERF_PROF: per-phase cycle stamps, REFNERF_LDS_PAD: force one
 *    workgroup per CU) and the 2 KB device buffer of the stamps, allocated on first use and kept for the process. */
struct Runtime {
  std::mutex mu;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<int> family;       /* REFNERF_TIMER_* of each recorded pair */
  size_t events_used = 0;
  bool prof = false;
  int lds_pad = 0;
  long long *d_prof = nullptr;
  Runtime() {
    prof = getenv("REFNERF_PROF") != nullptr;
    if (const char *pad = getenv("REFNERF_LDS_PAD")) lds_pad = atoi(pad);
  }
};
Runtime &rt() {
  static Runtime r;
  return r;
}
/* cycle-stamp buffer of the REFNERF_PROF debug mode (8 waves x 32 slots) */
int prof_buffer(long long **out) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.mu);
  if (!r.d_prof) HIP_TRY(hipMalloc(&r.d_prof, 8 * 32 * sizeof(long long)));
  *out = r.d_prof;
  return REFNERF_OK;
}
/* begin / end of a timed launch on stream `st`; `slot` < 0 = timer off */
int timer_begin(hipStream_t st, long *slot, int fam = REFNERF_TIMER_FORWARD) {
  Runtime &r = rt();
  *slot = -1;
  std::lock_guard<std::mutex> lk(r.mu);
  if (!r.timing || r.events_used >= 65536) return REFNERF_OK;
  if (r.events_used == r.events.size()) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    r.events.emplace_back(e0, e1);
    r.family.push_back(fam);
  }
  *slot = (long)r.events_used++;
  r.family[*slot] = fam;
  HIP_TRY(hipEventRecord(r.events[*slot].first, st));
  return REFNERF_OK;
}
int timer_end(hipStream_t st, long slot) {
  if (slot < 0) return REFNERF_OK;
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.mu);
  HIP_TRY(hipEventRecord(r.events[slot].second, st));
  return REFNERF_OK;
}
/* opt a kernel into the 160 KiB dynamic-LDS limit; the first failure is remembered and reported by every later call */
template <typename K>
hipError_t lds_attr(K kernel, int bytes = 160 * 1024) {
  return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
#define LDS_ATTR_ONCE(...)                                                                         \
  do {                                                                                             \
    static hipError_t attr_err_ = [] {                                                             \
      hipError_t es_[] = {__VA_ARGS__};                                                            \
      for (hipError_t e_ : es_) if (e_ != hipSuccess) return e_;                                   \
      return hipSuccess;                                                                           \
    }();                                                                                           \
    if (attr_err_ != hipSuccess) return fail(REFNERF_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr_err_)); \
  } while (0)

/* REFNERF_LEGACY_F16X2_TRAIN (debug knob, read once): training levels in REFNERF_PREC_F16X2 on the built-in basis take the round-4
 * kernels (fp32 skeleton with split-f16 chains, REFNERF_ACT_F16X2, d_packed = the REFNERF_PREC_F32 image) instead of the
 * round-5 ones (refnerf_sq_train.hip, REFNERF_ACT_SQ, d_packed = the REFNERF_IMAGE_F16X2_TRAIN image) */
bool legacy_f16x2_train() {
  static const bool v = [] { const char *e = getenv("REFNERF_LEGACY_F16X2_TRAIN"); return e && *e && *e != '0'; }();
  return v;
}

}  // namespace

namespace rnh {
int fail(int code, const char *fmt, const char *detail) { return ::fail(code, fmt, detail); }
int timer_begin(hipStream_t st, long *slot, int fam) { return ::timer_begin(st, slot, fam); }
int timer_end(hipStream_t st, long slot) { return ::timer_end(st, slot); }
bool prof_on() { return rt().prof; }
int prof_buffer(long long **out) { return ::prof_buffer(out); }
int lds_pad() { return rt().lds_pad; }
}  // namespace rnh

extern "C" {

int refnerf_abi_version(void) { return REFNERF_ABI_VERSION; }
const char *refnerf_last_error(void) { return g_err; }

void refnerf_level_cfg_default(refnerf_level_cfg *c) {
  memset(c, 0, sizeof(*c));
  c->n_samples = 128; c->n_in = 1; c->training = 0; c->compute_extras = 1;
  c->srgb_mapping = 1; c->srgb_mapping_normalization = 1; c->render_srgb_mode = REFNERF_SRGB_NONE;
  c->opaque_background = 0; c->ray_shape = 0; c->precision = REFNERF_PREC_F32; c->wgrad_mode = REFNERF_WGRAD_BF16X3; c->dir_enc = REFNERF_DIRENC_IDE; c->raydist = REFNERF_RAYDIST_NONE; c->disable_integration = 0; c->ipe_groups = 0;
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
  if (precision == REFNERF_PREC_BF16 || precision == REFNERF_PREC_F16) return ((size_t)rn::BFPACKED.chunks_per_pass + 2) * rn::BF_CHUNK_BYTES;
  if (precision == REFNERF_PREC_F16X2) return ((size_t)rn::SPPACKED.total_chunks + 2) * rn::BF_CHUNK_BYTES;
  if (precision == REFNERF_IMAGE_F16X2_TRAIN) return rnsq::image_bytes();
  return 0;
}

/* ---- which image a device pointer holds (v11): recorded by the pack entries, checked by the level entries ---- */
#include <mutex>
#include <unordered_map>
namespace {
constexpr int IMAGE_KIND_BASIS = 0x100;      /* + REFNERF_PREC_F32: the extended image of refnerf_pack_weights_basis */
std::mutex g_image_mu;
std::unordered_map<const void *, int> g_image_kind;
void remember_image(const void *p, int kind) {
  std::lock_guard<std::mutex> lk(g_image_mu);
  g_image_kind[p] = kind;
}
const char *image_name(int kind) {
  switch (kind) {
    case REFNERF_PREC_F32: return "REFNERF_PREC_F32";
    case REFNERF_PREC_BF16: return "REFNERF_PREC_BF16";
    case REFNERF_PREC_F16: return "REFNERF_PREC_F16";
    case REFNERF_PREC_F16X2: return "REFNERF_PREC_F16X2";
    case REFNERF_IMAGE_F16X2_TRAIN: return "REFNERF_IMAGE_F16X2_TRAIN";
    case IMAGE_KIND_BASIS + REFNERF_PREC_F32: return "REFNERF_PREC_F32 (general basis)";
    default: return "?";
  }
}
/* 0, or REFNERF_EINVAL when the library itself packed `p` as something else than this level streams */
int check_image(const void *p, const refnerf_level_cfg *cfg, const char *who) {
  const int want = refnerf_level_image(cfg) + (cfg->ipe_groups > 1 ? IMAGE_KIND_BASIS : 0);
  int have = -1;
  {
    std::lock_guard<std::mutex> lk(g_image_mu);
    auto it = g_image_kind.find(p);
    if (it != g_image_kind.end()) have = it->second;
  }
  if (have < 0 || have == want) return REFNERF_OK;
  /* (the plain f32 image in place of the extended one reads past its end; the extended one in place of the plain one is a superset) */
  if (have == IMAGE_KIND_BASIS + REFNERF_PREC_F32 && want == REFNERF_PREC_F32) return REFNERF_OK;
  snprintf(g_err, sizeof(g_err), "%s: d_packed was packed as the %s image, this level configuration streams the %s image (refnerf_level_image)",
           who, image_name(have), image_name(want));
  return REFNERF_EINVAL;
}
}  // namespace

int refnerf_level_image(const refnerf_level_cfg *cfg) {
  if (!cfg) return -1;
  if (cfg->ipe_groups > 1) return REFNERF_PREC_F32;
  if (cfg->training) return (cfg->precision == REFNERF_PREC_F16X2 && !legacy_f16x2_train()) ? REFNERF_IMAGE_F16X2_TRAIN : REFNERF_PREC_F32;
  return cfg->precision;
}

int refnerf_pack_weights(const float *d_params, void *d_packed, int precision, void *stream) {
  if (!d_params || !d_packed) return fail(REFNERF_EINVAL, "refnerf_pack_weights: null pointer%s");
  remember_image(d_packed, precision);
  if (precision == REFNERF_PREC_F32) {
    dim3 grid(64, 3 * rn::NUM_OPS + 3 * rn::NUM_TOPS + 1);
    hipLaunchKernelGGL(rn::pack_weights_f32, grid, dim3(256), 0, (hipStream_t)stream, d_params, (float *)d_packed);
  } else if (precision == REFNERF_PREC_BF16) {
    dim3 grid(8, rn::NUM_OPS);
    hipLaunchKernelGGL(rn::pack_weights_16<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, d_params, (char *)d_packed);
  } else if (precision == REFNERF_PREC_F16) {
    dim3 grid(8, rn::NUM_OPS);
    hipLaunchKernelGGL(rn::pack_weights_16<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, d_params, (char *)d_packed);
  } else if (precision == REFNERF_PREC_F16X2) {
    dim3 grid(8, rn::NUM_OPS);
    hipLaunchKernelGGL(rn::pack_weights_split, grid, dim3(256), 0, (hipStream_t)stream, d_params, (char *)d_packed);
  } else if (precision == REFNERF_IMAGE_F16X2_TRAIN) {
    return rnsq::pack(d_params, d_packed, (hipStream_t)stream);
  } else {
    return fail(REFNERF_EINVAL, "refnerf_pack_weights: unknown precision%s");
  }
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

size_t refnerf_packed_weights_bytes_basis(int precision, int ipe_groups) {
  if (ipe_groups <= 1) return refnerf_packed_weights_bytes(precision);
  if (precision != REFNERF_PREC_F32 || ipe_groups > rn::IPE_MAX_GROUPS) return 0;
  return (size_t)rn::PACKED_EXT_TOTAL * sizeof(float);
}

int refnerf_pack_weights_basis(const float *d_params, const float *d_basis, int ipe_groups, void *d_packed, int precision, void *stream) {
  if (ipe_groups <= 1) return refnerf_pack_weights(d_params, d_packed, precision, stream);
  if (!d_basis) return fail(REFNERF_EINVAL, "refnerf_pack_weights_basis: null basis%s");
  if (ipe_groups > rn::IPE_MAX_GROUPS) return fail(REFNERF_EUNSUPPORTED, "refnerf_pack_weights_basis: at most 7 groups of three directions (21: icosahedron / 2)%s");
  if (precision != REFNERF_PREC_F32)
    return fail(REFNERF_EUNSUPPORTED, "refnerf_pack_weights_basis builds the REFNERF_PREC_F32 image (levels with REFNERF_PREC_F32 or REFNERF_PREC_F16X2 run on it); the plain bf16 / f16 images have no direction groups%s");
  const int rc = refnerf_pack_weights(d_params, d_packed, precision, stream);
  if (rc) return rc;
  remember_image(d_packed, IMAGE_KIND_BASIS + REFNERF_PREC_F32);
  hipLaunchKernelGGL(rn::pack_weights_ext, dim3(32, 1 + 8 * rn::EXT_GROUPS), dim3(256), 0, (hipStream_t)stream, d_params, d_basis, ipe_groups, (float *)d_packed);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

static int rays_per_wg(int N, int tile) {
  if (N % tile == 0) return 1;
  if (tile % N == 0) return tile / N;
  /* otherwise whole passes only if a small multiple fits the per-sample LDS budget */
  if ((2 * N) % tile == 0 && 2 * N <= 512) return 2;
  if ((4 * N) % tile == 0 && 4 * N <= 512) return 4;
  return 1;   /* last pass partially filled */
}

namespace {
struct BwdPlan { long long S, pitch; int slices, k_per_slice; size_t act_bytes, delta_off, part_off, seed_off, cmin_off, total, act_ext_off, part_ext_off; };
/* groups > 1 (general IPE basis): the tail matrix of the groups' IPE features behind ACT, the tail's split-K partials behind the seeds */
/* split-K slices of the weight-gradient GEMMs (one PART image of NUM_PARAMS floats each, reduced in a fixed order) */
#ifndef REFNERF_MAX_SLICES
#define REFNERF_MAX_SLICES 32
#endif
BwdPlan bwd_plan(int R, int N, int groups = 0) {
  BwdPlan p;
  p.S = (long long)R * N;
  p.pitch = (p.S + 127) / 128 * 128;
  long long sl = (p.S + 2047) / 2048;
  p.slices = (int)(sl < 1 ? 1 : (sl > REFNERF_MAX_SLICES ? REFNERF_MAX_SLICES : sl));
  long long per = (p.S + p.slices - 1) / p.slices;
  p.k_per_slice = (int)((per + rn::WG_KT - 1) / rn::WG_KT * rn::WG_KT);
  p.act_bytes = sizeof(float) * (size_t)rn::ACT_ALLOC_ROWS * p.pitch;
  p.delta_off = 0;
  p.part_off = p.delta_off + sizeof(float) * (size_t)rn::DEL_ALLOC_ROWS * p.pitch;
  p.seed_off = p.part_off + sizeof(float) * (size_t)p.slices * rn::NUM_PARAMS;
  p.cmin_off = p.seed_off + sizeof(float) * (size_t)rn::NGS * p.pitch;      /* split-f16 formats: the smallest factor per layer id */
  p.total = p.cmin_off + 128;
  p.act_ext_off = p.act_bytes;
  p.part_ext_off = p.total;
  if (groups > 1) {
    p.act_bytes += sizeof(float) * (size_t)rn::ACT_EXT_UNITS * p.pitch;
    p.total += sizeof(float) * (size_t)p.slices * rn::EXT_PARAMS;
  }
  return p;
}
}  // namespace

static int level_forward_impl(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays,
                              int32_t R, const float *d_sdist_in, const float *d_weights_in,
                              const refnerf_level_out *out, float *d_act, long long act_pitch, void *stream) {
  if (!d_packed || !cfg || !rays || !out || !d_sdist_in || !d_weights_in)
    return fail(REFNERF_EINVAL, "refnerf_level_forward: null pointer%s");
  if (R <= 0) return fail(REFNERF_EINVAL, "refnerf_level_forward: R must be positive%s");
  /* stepfun.py:234-235 */
  if (cfg->n_samples <= 1) return fail(REFNERF_EINVAL, "num_samples must be > 1%s");
  /* render.py:126 */
  if (cfg->ray_shape != 0 && cfg->ray_shape != 1) return fail(REFNERF_EINVAL, "ray_shape must be 'cone' or 'cylinder'%s");
  if (cfg->n_in < 1 || cfg->n_in > 512) return fail(REFNERF_EINVAL, "n_in must be in [1,512]%s");
  if (cfg->precision != REFNERF_PREC_F32 && cfg->precision != REFNERF_PREC_BF16 && cfg->precision != REFNERF_PREC_F16 &&
      cfg->precision != REFNERF_PREC_F16X2)
    return fail(REFNERF_EINVAL, "unknown precision mode%s");
  if (cfg->raydist < REFNERF_RAYDIST_NONE || cfg->raydist > REFNERF_RAYDIST_SQUARE)
    return fail(REFNERF_EINVAL, "unknown raydist (REFNERF_RAYDIST_*)%s");
  if (cfg->dir_enc != REFNERF_DIRENC_IDE && cfg->dir_enc != REFNERF_DIRENC_POSENC)
    return fail(REFNERF_EINVAL, "unknown dir_enc (REFNERF_DIRENC_IDE / REFNERF_DIRENC_POSENC)%s");
  if (cfg->training && cfg->precision == REFNERF_PREC_F16)
    return fail(REFNERF_EUNSUPPORTED, "REFNERF_PREC_F16 is an inference mode (training levels: REFNERF_PREC_F32, REFNERF_PREC_F16X2 or REFNERF_PREC_BF16)%s");
  const bool gbasis = cfg->ipe_groups > 1;
  if (cfg->ipe_groups < 0 || cfg->ipe_groups > rn::IPE_MAX_GROUPS) return fail(REFNERF_EINVAL, "ipe_groups must be in [0,7]%s");
  if (gbasis && cfg->precision != REFNERF_PREC_F32 && cfg->precision != REFNERF_PREC_F16X2)
    return fail(REFNERF_EUNSUPPORTED, "a general IPE basis (ipe_groups > 1) runs with REFNERF_PREC_F32 or REFNERF_PREC_F16X2 (d_packed: the REFNERF_PREC_F32 image of refnerf_pack_weights_basis in both)%s");
  /* training + BF16: the fp32-structure kernel with its MLP chains on bf16 MFMA (level_fwd_train_bf16c); d_packed is
   * the REFNERF_PREC_F32 image in that case (it carries the bf16 copies of the ops) */
  const bool train_bf = cfg->training && cfg->precision == REFNERF_PREC_BF16;
  if (!rays->d_origins || !rays->d_directions || !rays->d_viewdirs || !rays->d_radii || !rays->d_near || !rays->d_far)
    return fail(REFNERF_EINVAL, "refnerf_level_forward: null ray field%s");
  const int N = cfg->n_samples;
  /* training + F16X2: the same kernel with its chains on split-f16 operands (level_fwd_train_f16x2c), fp32 ACT rows */
  /* (a general basis in F16X2 takes that kernel in inference as well: level_fwd_f16x2c_gb) */
  const bool gb_split = gbasis && cfg->precision == REFNERF_PREC_F16X2;
  /* training + F16X2 on the built-in basis: the round-5 kernels on the eval kernel's skeleton (refnerf_sq_train.hip);
   * d_packed is the REFNERF_IMAGE_F16X2_TRAIN image, the activations REFNERF_ACT_SQ */
  if (int irc = check_image(d_packed, cfg, "refnerf_level_forward")) return irc;
  if (cfg->training && cfg->precision == REFNERF_PREC_F16X2 && !gbasis && !d_act && !legacy_f16x2_train())
    return fail(REFNERF_EINVAL, "a training level in REFNERF_PREC_F16X2 runs through refnerf_level_forward_train: its kernel keeps the ReLU sign words and "
                                "the bottleneck rows in the activation buffer (d_packed: the REFNERF_IMAGE_F16X2_TRAIN image)%s");
  if (cfg->wgrad_mode == REFNERF_WGRAD_F16 && !(cfg->training && cfg->precision == REFNERF_PREC_F16X2 && !gbasis && !legacy_f16x2_train()))
    return fail(REFNERF_EUNSUPPORTED, "wgrad_mode = REFNERF_WGRAD_F16 belongs to training levels in REFNERF_PREC_F16X2 on the built-in IPE basis%s");
  if (cfg->training && cfg->precision == REFNERF_PREC_F16X2 && !gbasis && d_act && !legacy_f16x2_train())
    return rnsq::forward(d_packed, cfg, rays, R, d_sdist_in, d_weights_in, out, d_act, (hipStream_t)stream);
  const bool train_split = (cfg->training && cfg->precision == REFNERF_PREC_F16X2) || gb_split;
  const bool split = cfg->precision == REFNERF_PREC_F16X2 && !train_split;
  const bool bf = (cfg->precision == REFNERF_PREC_BF16 && !train_bf) || cfg->precision == REFNERF_PREC_F16 || split;     /* the LDS-ring 16-bit eval kernels */
  int rpw = rays_per_wg(N, bf ? rn::BT : rn::T_TILE);
  /* 16-bit inference, rays that do not tile the 256-sample pass within 640 samples (N = 192: 2 rays = one and a half
   * passes): take the smallest ray count that does (N = 192: 4 rays = three full passes) with the per-sample records in a
   * ring (level_fwd_*_ring composites every ray behind the pass that completes it) -- if a ray plus a pass fit the ring,
   * there are still at least two workgroups per CU, and the per-ray arrays fit */
  bool ps_ring = false;
  if (bf && N <= rn::BF_PS_RING - rn::BT && (rpw * N) % rn::BT != 0) {
    int r = 1;
    while (r <= rn::BF_NW && (r * N) % rn::BT != 0) ++r;
    if (r <= rn::BF_NW && r * N > 640 && R / r >= 512) { rpw = r; ps_ring = true; }
  }
  auto lds_bytes = [&](int rays) -> size_t {
    const int np = bf ? rn::NPS_EVAL : rn::NPS_TRAIN, tile = bf ? rn::BT : rn::T_TILE;
    const int ps_rows = ps_ring ? rn::BF_PS_RING : rays * N;
    const size_t per_wg = sizeof(float) * (size_t)(2 * rays * (N + 1) + np * ps_rows + 3 * tile + 8 + 12 * rays);
    if (bf) return (size_t)rn::BF_RING_BYTES + rn::BF_X_BYTES + sizeof(float) * rn::HD_ROWS * rn::BT + per_wg;
    return sizeof(float) * (size_t)(rn::DIR_PAD * rn::T_TILE + rn::HD_ROWS * rn::T_TILE) + per_wg;
  };
  if (ps_ring && lds_bytes(rpw) + (size_t)rt().lds_pad > 160 * 1024) { ps_ring = false; rpw = rays_per_wg(N, rn::BT); }
  /* bf16: the per-ray phases (resample, compositing) occupy one wave per ray, so take as many rays per
   * workgroup as the LDS holds (up to one per wave): the other waves idle for a shorter share of the pass */
  if (bf && !ps_ring) while (2 * rpw <= rn::BF_NW && 2 * rpw * N <= 640 && lds_bytes(2 * rpw) <= 160 * 1024 &&
                 R / (2 * rpw) >= 512)        /* ... but keep at least two workgroups per CU in flight */
    rpw *= 2;
  if (rpw * N > 640 && !ps_ring) return fail(REFNERF_EINVAL, "n_samples too large for the LDS budget (rays_per_wg*N must be <= 640)%s");
  size_t lds = lds_bytes(rpw);
  {
    const int nwmax = bf ? rn::BF_NW : 4;
    const int nw = rpw < nwmax ? rpw : nwmax;
    const size_t scratch = sizeof(float) * (size_t)nw * (3 * (cfg->n_in + 4) + N + 3);
    if (scratch > (bf ? (size_t)rn::BF_X_BYTES : sizeof(float) * rn::DIR_PAD * rn::T_TILE))
      return fail(REFNERF_EINVAL, "n_in / n_samples too large for the resampler scratch of this precision mode%s");
  }
  /* bf16 chains: the shared weight-stream ring comes on top; fewer rays per workgroup (a partly filled last pass)
   * when the whole-pass choice no longer fits */
  /* ... and so do the split-f16 chains of the built-in basis (REFNERF_SPLIT_SHARED: half steps through the same 8 KB slots) */
  const bool train_ring = train_bf || (train_split && !gbasis && REFNERF_SPLIT_SHARED != 0);
  while (train_ring && rpw > 1 && (lds + 15) / 16 * 16 + rn::RING_BYTES > 160 * 1024) { rpw /= 2; lds = lds_bytes(rpw); }
  const size_t ring_off = (lds + 15) / 16 * 16;
  if (train_ring) lds = ring_off + rn::RING_BYTES;
  lds += (size_t)rt().lds_pad;   /* debug (REFNERF_LDS_PAD): force 1 workgroup/CU */
  if (lds > 160 * 1024) return fail(REFNERF_EINVAL, "n_samples too large for the 160 KiB LDS budget of this precision mode%s");
  LDS_ATTR_ONCE(lds_attr(rn::level_fwd_f32), lds_attr(rn::level_fwd_train_f32), lds_attr(rn::level_fwd_train_bf16c), lds_attr(rn::level_fwd_train_f16x2c),
                lds_attr(rn::level_fwd_bf16), lds_attr(rn::level_fwd_f16), lds_attr(rn::level_fwd_bf16_ring), lds_attr(rn::level_fwd_f16_ring),
                lds_attr(rn::level_fwd_f16x2), lds_attr(rn::level_fwd_f16x2_ring), lds_attr(rn::level_fwd_f32_gb), lds_attr(rn::level_fwd_train_f32_gb), lds_attr(rn::level_fwd_f16x2c_gb));
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
  a.act = d_act; a.act_pitch = act_pitch;
  a.ring_off = (int)ring_off;
  if (rt().prof) {
    int prc = prof_buffer(&a.prof);
    if (prc) return prc;
  }
  int grid = (R + rpw - 1) / rpw;
  hipStream_t st = (hipStream_t)stream;
  long tslot = -1;
  {
    int trc = timer_begin(st, &tslot);
    if (trc) return trc;
  }
  if (split && ps_ring) hipLaunchKernelGGL(rn::level_fwd_f16x2_ring, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (split) hipLaunchKernelGGL(rn::level_fwd_f16x2, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (bf && ps_ring && cfg->precision == REFNERF_PREC_F16) hipLaunchKernelGGL(rn::level_fwd_f16_ring, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (bf && ps_ring) hipLaunchKernelGGL(rn::level_fwd_bf16_ring, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (bf && cfg->precision == REFNERF_PREC_F16) hipLaunchKernelGGL(rn::level_fwd_f16, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (bf) hipLaunchKernelGGL(rn::level_fwd_bf16, dim3(grid), dim3(rn::BF_NTHREADS), lds, st, a);
  else if (train_bf) hipLaunchKernelGGL(rn::level_fwd_train_bf16c, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else if (gb_split) hipLaunchKernelGGL(rn::level_fwd_f16x2c_gb, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else if (train_split) hipLaunchKernelGGL(rn::level_fwd_train_f16x2c, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else if (cfg->training && gbasis) hipLaunchKernelGGL(rn::level_fwd_train_f32_gb, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else if (cfg->training) hipLaunchKernelGGL(rn::level_fwd_train_f32, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else if (gbasis) hipLaunchKernelGGL(rn::level_fwd_f32_gb, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  else hipLaunchKernelGGL(rn::level_fwd_f32, dim3(grid), dim3(rn::NTHREADS), lds, st, a);
  HIP_TRY(hipGetLastError());
  {
    int trc = timer_end(st, tslot);
    if (trc) return trc;
  }
  if (a.prof) {   /* debug aid (REFNERF_PROF=1): per-phase cycle stamps of a mid-grid workgroup */
    long long hbuf[8 * 32];
    HIP_TRY(hipMemcpy(hbuf, a.prof, sizeof(hbuf), hipMemcpyDeviceToHost));
    const int nw = bf ? 8 : 4;
    for (int w = 0; w < nw; ++w) {
      fprintf(stderr, "[prof] wave %d:", w);
      for (int sl = 1; sl <= 18; ++sl) fprintf(stderr, " %lld", hbuf[w * 32 + sl] ? hbuf[w * 32 + sl] - hbuf[w * 32] : -1LL);
      fprintf(stderr, "  | dma-wait %lld barrier-wait %lld", hbuf[w * 32 + 20], hbuf[w * 32 + 21]);
      if (hbuf[w * 32 + 23] > hbuf[w * 32 + 22])   /* -DREFNERF_PROF_WAITS builds: shader clock from the 100 MHz counter */
        fprintf(stderr, " | clock %.0f MHz", 100.0 * (double)(hbuf[w * 32 + 15] - hbuf[w * 32]) / (double)(hbuf[w * 32 + 23] - hbuf[w * 32 + 22]));
      fprintf(stderr, "\n");
    }
  }
  return REFNERF_OK;
}

int refnerf_level_forward(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays,
                          int32_t R, const float *d_sdist_in, const float *d_weights_in,
                          const refnerf_level_out *out, void *stream) {
  return level_forward_impl(d_packed, cfg, rays, R, d_sdist_in, d_weights_in, out, nullptr, 0, stream);
}

int refnerf_level_forward_train(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays,
                                int32_t R, const float *d_sdist_in, const float *d_weights_in,
                                const refnerf_level_out *out, void *d_activations, size_t activations_bytes,
                                void *stream) {
  if (!cfg || !d_activations) return fail(REFNERF_EINVAL, "refnerf_level_forward_train: null pointer%s");
  if (!cfg->training) return fail(REFNERF_EINVAL, "refnerf_level_forward_train: cfg->training must be 1%s");
  if (R <= 0 || cfg->n_samples <= 1) return fail(REFNERF_EINVAL, "refnerf_level_forward_train: bad R / num_samples%s");
  const BwdPlan plan = bwd_plan(R, cfg->n_samples, cfg->ipe_groups);
  if (activations_bytes < plan.act_bytes)
    return fail(REFNERF_EINVAL, "refnerf_level_forward_train: activation buffer too small (see refnerf_activation_workspace_bytes)%s");
  return level_forward_impl(d_packed, cfg, rays, R, d_sdist_in, d_weights_in, out, (float *)d_activations, plan.pitch, stream);
}

int refnerf_pixels_to_rays(const int32_t *d_pix_x, const int32_t *d_pix_y, const float *d_pixtocams, int32_t pixtocam_per_ray,
                           const float *d_camtoworlds, int32_t camtoworld_per_ray, const float *d_pixtocam_ndc, int32_t n,
                           float *d_origins, float *d_directions, float *d_viewdirs, float *d_radii, float *d_imageplane,
                           void *stream) {
  if (!d_pix_x || !d_pix_y || !d_pixtocams || !d_camtoworlds || !d_origins || !d_directions || !d_viewdirs || !d_radii)
    return fail(REFNERF_EINVAL, "refnerf_pixels_to_rays: null pointer%s");
  if (n <= 0) return fail(REFNERF_EINVAL, "refnerf_pixels_to_rays: n must be positive%s");
  rn::RayGenArgs a;
  a.pix_x = d_pix_x; a.pix_y = d_pix_y;
  a.pixtocams = d_pixtocams; a.camtoworlds = d_camtoworlds; a.pixtocam_ndc = d_pixtocam_ndc;
  a.p2c_stride = pixtocam_per_ray ? 9 : 0;
  a.c2w_stride = camtoworld_per_ray ? 12 : 0;
  a.n = n;
  a.origins = d_origins; a.directions = d_directions; a.viewdirs = d_viewdirs; a.radii = d_radii; a.imageplane = d_imageplane;
  hipLaunchKernelGGL(rn::pixels_to_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_mlp_forward(const void *d_packed, const refnerf_level_cfg *cfg, const float *d_means, const float *d_covs,
                        int32_t cov_is_full, const float *d_viewdirs, int32_t R, int32_t N, const refnerf_level_out *out,
                        void *stream) {
  if (!d_packed || !cfg || !d_means || !d_covs || !d_viewdirs || !out)
    return fail(REFNERF_EINVAL, "refnerf_mlp_forward: null pointer%s");
  if (R <= 0 || N <= 0) return fail(REFNERF_EINVAL, "refnerf_mlp_forward: R and N must be positive%s");
  if (cfg->precision != REFNERF_PREC_F32)
    return fail(REFNERF_EUNSUPPORTED, "refnerf_mlp_forward runs in the f32 precision mode only%s");
  int rpw = rays_per_wg(N, rn::T_TILE);
  if (rpw * N > 640) return fail(REFNERF_EINVAL, "n too large for the LDS budget (rays_per_wg*N must be <= 640)%s");
  const size_t lds = sizeof(float) * (size_t)(rn::DIR_PAD * rn::T_TILE + rn::HD_ROWS * rn::T_TILE + 2 * rpw * (N + 1) +
                                              rn::NPS_TRAIN * rpw * N + 3 * rn::T_TILE + 8);
  if (lds > 160 * 1024) return fail(REFNERF_EINVAL, "n too large for the 160 KiB LDS budget%s");
  if (cfg->ipe_groups < 0 || cfg->ipe_groups > rn::IPE_MAX_GROUPS) return fail(REFNERF_EINVAL, "ipe_groups must be in [0,7]%s");
  const bool gbasis = cfg->ipe_groups > 1;     /* d_packed then comes from refnerf_pack_weights_basis */
  LDS_ATTR_ONCE(lds_attr(rn::mlp_fwd_f32), lds_attr(rn::mlp_fwd_train_f32), lds_attr(rn::mlp_fwd_f32_gb), lds_attr(rn::mlp_fwd_train_f32_gb));
  rn::LevelArgs a{};
  a.packed = d_packed;
  a.cfg = *cfg;
  a.cfg.n_samples = N;
  a.rays.d_viewdirs = d_viewdirs;
  a.R = R;
  a.rpw = rpw;
  a.out = *out;
  a.prof = nullptr;
  a.g_means = d_means;
  a.g_covs = d_covs;
  a.cov_full = cov_is_full;
  const int grid = (R + rpw - 1) / rpw;
  if (gbasis && cfg->training) hipLaunchKernelGGL(rn::mlp_fwd_train_f32_gb, dim3(grid), dim3(rn::NTHREADS), lds, (hipStream_t)stream, a);
  else if (gbasis) hipLaunchKernelGGL(rn::mlp_fwd_f32_gb, dim3(grid), dim3(rn::NTHREADS), lds, (hipStream_t)stream, a);
  else if (cfg->training) hipLaunchKernelGGL(rn::mlp_fwd_train_f32, dim3(grid), dim3(rn::NTHREADS), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(rn::mlp_fwd_f32, dim3(grid), dim3(rn::NTHREADS), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}


size_t refnerf_backward_workspace_bytes(int32_t R, int32_t n_samples) {
  if (R <= 0 || n_samples <= 1) return 0;
  return bwd_plan(R, n_samples).total;
}

size_t refnerf_activation_workspace_bytes(int32_t R, int32_t n_samples) {
  if (R <= 0 || n_samples <= 1) return 0;
  return bwd_plan(R, n_samples).act_bytes;
}

size_t refnerf_backward_workspace_bytes_basis(int32_t R, int32_t n_samples, int32_t ipe_groups) {
  if (R <= 0 || n_samples <= 1 || ipe_groups < 0 || ipe_groups > rn::IPE_MAX_GROUPS) return 0;
  return bwd_plan(R, n_samples, ipe_groups).total;
}

size_t refnerf_activation_workspace_bytes_basis(int32_t R, int32_t n_samples, int32_t ipe_groups) {
  if (R <= 0 || n_samples <= 1 || ipe_groups < 0 || ipe_groups > rn::IPE_MAX_GROUPS) return 0;
  return bwd_plan(R, n_samples, ipe_groups).act_bytes;
}

int refnerf_activations_format(const refnerf_level_cfg *cfg) {
  if (!cfg) return -1;
  if (cfg->precision == REFNERF_PREC_BF16) return REFNERF_ACT_BF16;
  if (cfg->precision == REFNERF_PREC_F16X2 && cfg->ipe_groups <= 1) return legacy_f16x2_train() ? REFNERF_ACT_F16X2 : REFNERF_ACT_SQ;
  return REFNERF_ACT_F32;
}

int refnerf_level_backward(const void *d_packed, const refnerf_level_cfg *cfg, const refnerf_rays *rays, int32_t R,
                           const refnerf_level_saved *saved, const refnerf_level_grads *grads, float *d_param_grads,
                           void *d_workspace, size_t workspace_bytes, void *stream) {
  if (!d_packed || !cfg || !rays || !saved || !grads || !d_param_grads || !d_workspace)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: null pointer%s");
  if (R <= 0) return fail(REFNERF_EINVAL, "refnerf_level_backward: R must be positive%s");
  if (cfg->n_samples <= 1) return fail(REFNERF_EINVAL, "num_samples must be > 1%s");
  if (cfg->ray_shape != 0 && cfg->ray_shape != 1) return fail(REFNERF_EINVAL, "ray_shape must be 'cone' or 'cylinder'%s");
  if (cfg->precision != REFNERF_PREC_F32 && cfg->precision != REFNERF_PREC_BF16 && cfg->precision != REFNERF_PREC_F16X2)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: unknown precision mode (REFNERF_PREC_F32, REFNERF_PREC_F16X2 or REFNERF_PREC_BF16)%s");
  if (cfg->precision == REFNERF_PREC_F16X2 && saved->activations_format != REFNERF_ACT_F32 && saved->activations_format != REFNERF_ACT_F16X2 &&
      saved->activations_format != REFNERF_ACT_SQ)
    return fail(REFNERF_EUNSUPPORTED, "the split-f16 backward chains read the REFNERF_PREC_F16X2 forward's activations (REFNERF_ACT_SQ) or fp32 rows (REFNERF_PREC_F32, or a general IPE basis)%s");
  if (cfg->precision != REFNERF_PREC_F16X2 && (saved->activations_format == REFNERF_ACT_F16X2 || saved->activations_format == REFNERF_ACT_SQ))
    return fail(REFNERF_EUNSUPPORTED, "activations written by the split-f16 training forward (REFNERF_ACT_F16X2) are read by the split-f16 backward: cfg->precision = REFNERF_PREC_F16X2%s");
  if (cfg->wgrad_mode != REFNERF_WGRAD_F32 && cfg->wgrad_mode != REFNERF_WGRAD_BF16X3 && cfg->wgrad_mode != REFNERF_WGRAD_F16)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: unknown wgrad_mode%s");
  if (cfg->wgrad_mode == REFNERF_WGRAD_F16 && saved->activations_format != REFNERF_ACT_SQ)
    return fail(REFNERF_EUNSUPPORTED, "wgrad_mode = REFNERF_WGRAD_F16 belongs to the REFNERF_PREC_F16X2 training kernels (REFNERF_ACT_SQ activations, written with the same wgrad_mode)%s");
  {
    /* the backward streams the image of the TRAINING level it belongs to, whatever cfg->training says */
    refnerf_level_cfg tc = *cfg;
    tc.training = 1;
    /* (the split-f16 backward on fp32 rows -- an exact-fp32 forward, or a general basis -- reads the f32 image) */
    if (tc.precision == REFNERF_PREC_F16X2 && saved->activations_format == REFNERF_ACT_F32) tc.precision = REFNERF_PREC_F32;
    if (int irc = check_image(d_packed, &tc, "refnerf_level_backward")) return irc;
  }
  if (!saved->d_sdist || !saved->d_density || !saved->d_rgb || !saved->d_weights || !grads->d_g_r_rgb)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: null saved tensor / rendering gradient%s");
  if (!saved->d_activations)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: the level was not run through refnerf_level_forward_train (no saved activations)%s");
  if (!rays->d_origins || !rays->d_directions || !rays->d_viewdirs || !rays->d_radii || !rays->d_near || !rays->d_far)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: null ray field%s");
  const int N = cfg->n_samples;
  const bool gbasis = cfg->ipe_groups > 1;
  if (cfg->ipe_groups < 0 || cfg->ipe_groups > rn::IPE_MAX_GROUPS) return fail(REFNERF_EINVAL, "ipe_groups must be in [0,7]%s");
  if (gbasis && ((cfg->precision != REFNERF_PREC_F32 && cfg->precision != REFNERF_PREC_F16X2) || cfg->wgrad_mode != REFNERF_WGRAD_BF16X3 ||
                 saved->activations_format != REFNERF_ACT_F32))
    return fail(REFNERF_EUNSUPPORTED, "a general IPE basis (ipe_groups > 1) trains with the f32 or split-f16 chains and the bf16x3 weight-gradient GEMM%s");
  const BwdPlan plan = bwd_plan(R, N, cfg->ipe_groups);
  if (workspace_bytes < plan.total) return fail(REFNERF_EINVAL, "refnerf_level_backward: workspace too small (see refnerf_backward_workspace_bytes)%s");
  const int rpw = rays_per_wg(N, rn::T_TILE);
  size_t lds = sizeof(float) * (size_t)(rn::DIR_PAD * rn::T_TILE + rn::HD_ROWS * rn::T_TILE + rpw * (N + 1) + 8);
  const size_t ring_off = (lds + 15) / 16 * 16;
  if (cfg->precision == REFNERF_PREC_BF16 || (cfg->precision == REFNERF_PREC_F16X2 && saved->activations_format == REFNERF_ACT_F16X2 && REFNERF_SPLIT_SHARED != 0))
    lds = ring_off + rn::RING_BYTES;       /* the chains' shared weight-stream ring */
  if (lds > 160 * 1024) return fail(REFNERF_EINVAL, "n_samples too large for the 160 KiB LDS budget%s");
  LDS_ATTR_ONCE(lds_attr(rn::level_bwd_f32), lds_attr(rn::level_bwd_bf16c), lds_attr(rn::level_bwd_f16x2c), lds_attr(rn::level_bwd_f16x2c_r32),
                lds_attr(rn::wgrad_f16s_kernel<rn::WF_NW>, rn::wf_lds(rn::WF_NW)),
                lds_attr(rn::wgrad_kernel, (rn::WG_TM + rn::WG_TN) * rn::WG_LDK * 4),
                lds_attr(rn::wgrad_bf16x3_kernel<false, false>, rn::wb_lds(false, false)),
                lds_attr(rn::wgrad_bf16x3_kernel<false, true>, rn::wb_lds(false, true)),
                lds_attr(rn::wgrad_bf16x3_kernel<true, false>, rn::wb_lds(true, false)),
                lds_attr(rn::wgrad_bf16x3_kernel<true, true>, rn::wb_lds(true, true)),
                lds_attr(rn::wgrad_bf16x3_kernel<false, false, true>, rn::wb_lds(false, false)));
  char *ws = (char *)d_workspace;
  rn::BwdArgs a;
  a.packed = d_packed;
  a.cfg = *cfg;
  a.rays = *rays;
  a.R = R;
  a.rpw = rpw;
  a.sdist = saved->d_sdist; a.density = saved->d_density; a.rgb = saved->d_rgb; a.weights = saved->d_weights;
  a.g_r_rgb = grads->d_g_r_rgb; a.g_weights = grads->d_g_weights; a.g_npred = grads->d_g_normals_pred;
  a.g_r_acc = grads->d_g_r_acc; a.g_r_dist = grads->d_g_r_distance;
  a.g_s_density = grads->d_g_density; a.g_s_rgb = grads->d_g_rgb; a.g_s_diffuse = grads->d_g_diffuse;
  a.g_s_specular = grads->d_g_specular; a.g_s_tint = grads->d_g_tint; a.g_s_rough = grads->d_g_roughness;
  a.act = (const float *)saved->d_activations;
  a.delta = (float *)(ws + plan.delta_off);
  a.seeds = (float *)(ws + plan.seed_off);
  a.pitch = plan.pitch;
  const bool act16 = saved->activations_format == REFNERF_ACT_BF16, del16 = cfg->precision == REFNERF_PREC_BF16 && (REFNERF_DELTA16 != 0);
  /* split-f16 formats (refnerf_layout.h): ACT as hi / lo pair units, DELTA as one half per element + factor rows */
  const bool pairs = saved->activations_format == REFNERF_ACT_F16X2;
  const bool sq = saved->activations_format == REFNERF_ACT_SQ;
  if (saved->activations_format != REFNERF_ACT_F32 && saved->activations_format != REFNERF_ACT_BF16 && !pairs && !sq)
    return fail(REFNERF_EINVAL, "refnerf_level_backward: unknown activations_format%s");
  if (sq && gbasis) return fail(REFNERF_EUNSUPPORTED, "a general IPE basis keeps fp32 activation rows (REFNERF_ACT_F32)%s");
  if (sq && cfg->wgrad_mode != REFNERF_WGRAD_BF16X3 && cfg->wgrad_mode != REFNERF_WGRAD_F16)
    return fail(REFNERF_EUNSUPPORTED, "REFNERF_ACT_SQ activations go with wgrad_mode = REFNERF_WGRAD_BF16X3 (the f16 weight-gradient GEMM on the saved halves; "
                                      "fp32 weight-gradient products: the REFNERF_PREC_F32 chains)%s");
  if (pairs && gbasis) return fail(REFNERF_EUNSUPPORTED, "a general IPE basis keeps fp32 activation rows (REFNERF_ACT_F32)%s");
  if ((act16 || del16 || pairs) && cfg->wgrad_mode != REFNERF_WGRAD_BF16X3)
    return fail(REFNERF_EUNSUPPORTED, "16-bit activation / delta rows (bf16 chains, split-f16 pair units) need wgrad_mode = REFNERF_WGRAD_BF16X3 "
                                      "(fp32 weight-gradient products: the REFNERF_PREC_F32 chains)%s");
  a.act16 = act16 ? 1 : 0;
  a.ring_off = (int)ring_off;
  a.prof = nullptr;
  if (rt().prof) {
    int prc = prof_buffer(&a.prof);
    if (prc) return prc;
    HIP_TRY(hipMemset(a.prof, 0, 8 * 32 * sizeof(long long)));
  }
  hipStream_t st = (hipStream_t)stream;
  /* per-ray seeds first (one wave per ray), then the per-sample backward */
  hipLaunchKernelGGL(rn::bwd_seed_kernel, dim3((R + 3) / 4), dim3(rn::NTHREADS), sizeof(float) * 4 * (size_t)(N + 1), st, a);
  if (sq) {
    /* round-5 kernels: the per-sample chains, then the f16 weight-gradient GEMM on (ACT_SQ, DELTA + factor units) */
    int rc = rnsq::backward_chain(d_packed, cfg, rays, R, saved->d_sdist, grads, a.act, a.delta, a.seeds, plan.pitch, st);
    if (rc) return rc;
    if (plan.pitch > plan.S) {
      hipLaunchKernelGGL(rn::wgrad_zero_tail, dim3(256), dim3(256), 0, st, const_cast<float *>(a.act), rn::AQ_MASK, rn::AQ_UNITS, plan.pitch, plan.S);
      hipLaunchKernelGGL(rn::wgrad_zero_tail, dim3(256), dim3(256), 0, st, a.delta, rn::DEL_ROWS / 2, rn::DQ_UNITS, plan.pitch, plan.S);
    }
    const int slices = (int)((plan.S + plan.k_per_slice - 1) / plan.k_per_slice);
    float *part = (float *)(ws + plan.part_off);
    int used = slices;
    rc = rnsq::wgrad(a.act, a.delta, plan.S, plan.pitch, plan.k_per_slice, slices, part, (float *)(ws + plan.cmin_off),
                     cfg->wgrad_mode == REFNERF_WGRAD_F16 ? 1 : 0, &used, st);
    if (rc) return rc;
    hipLaunchKernelGGL(rn::wgrad_reduce, dim3(1024), dim3(256), 0, st, part, used, d_param_grads, (int)rn::NUM_PARAMS);
    HIP_TRY(hipGetLastError());
    return REFNERF_OK;
  }
  /* d_packed is the f32 image in both modes (it carries the bf16 transposed ops behind the fp32 ones) */
  long tslot = -1;
  { int trc = timer_begin(st, &tslot, REFNERF_TIMER_BACKWARD); if (trc) return trc; }
  if (cfg->precision == REFNERF_PREC_BF16)
    hipLaunchKernelGGL(rn::level_bwd_bf16c, dim3((R + rpw - 1) / rpw), dim3(rn::NTHREADS), lds, st, a);
  else if (cfg->precision == REFNERF_PREC_F16X2 && pairs)
    hipLaunchKernelGGL(rn::level_bwd_f16x2c, dim3((R + rpw - 1) / rpw), dim3(rn::NTHREADS), lds, st, a);
  else if (cfg->precision == REFNERF_PREC_F16X2)
    hipLaunchKernelGGL(rn::level_bwd_f16x2c_r32, dim3((R + rpw - 1) / rpw), dim3(rn::NTHREADS), lds, st, a);
  else
    hipLaunchKernelGGL(rn::level_bwd_f32, dim3((R + rpw - 1) / rpw), dim3(rn::NTHREADS), lds, st, a);
  HIP_TRY(hipGetLastError());
  { int trc = timer_end(st, tslot); if (trc) return trc; }
  if (a.prof) {   /* debug aid (REFNERF_PROF=1): cycle stamps of workgroup 0: prologue | heads recompute | rgb recompute + colour head | seed | dir chain | IDE / heads | spatial chain */
    long long hbuf[8 * 32];
    HIP_TRY(hipMemcpy(hbuf, a.prof, sizeof(hbuf), hipMemcpyDeviceToHost));
    for (int w = 0; w < 4; ++w) {
      fprintf(stderr, "[prof bwd] wave %d:", w);
      for (int sl = 1; sl <= 14; ++sl) fprintf(stderr, " %lld", hbuf[w * 32 + sl] ? hbuf[w * 32 + sl] - hbuf[w * 32] : -1LL);
      fprintf(stderr, "\n");
    }
  }
  if (plan.pitch > plan.S) {   /* pad columns of both operand matrices must read as zero in the wgrad GEMM */
    hipLaunchKernelGGL(rn::wgrad_zero_tail, dim3(256), dim3(256), 0, st, const_cast<float *>(a.act), act16 ? rn::ACT_ROWS / 2 : rn::ACT_ROWS, rn::act_units(act16), plan.pitch, plan.S);
    hipLaunchKernelGGL(rn::wgrad_zero_tail, dim3(256), dim3(256), 0, st, a.delta, (del16 || pairs) ? rn::DEL_ROWS / 2 : rn::DEL_ROWS,
                       pairs ? rn::DEL_UNITS_F16S : rn::del_units(del16), plan.pitch, plan.S);
  }
  rn::WgradArgs w;
  w.act = a.act; w.delta = a.delta; w.a_units = rn::act_units(act16); w.d_units = pairs ? rn::DEL_UNITS_F16S : rn::del_units(del16); w.pitch = plan.pitch; w.S = plan.S; w.k_per_slice = plan.k_per_slice;
  w.part = (float *)(ws + plan.part_off);
  const int slices = (int)((plan.S + plan.k_per_slice - 1) / plan.k_per_slice);
  { int trc = timer_begin(st, &tslot, REFNERF_TIMER_WGRAD); if (trc) return trc; }
  if (cfg->wgrad_mode == REFNERF_WGRAD_BF16X3)
  {
    const dim3 wg_grid(8 * ((slices + 7) / 8) * rn::WJOBS.tiles);
    if (pairs) {
      /* the layers' smallest factors first (18 exact minima: +inf bits, then one atomicMin per block), then the f16 GEMM */
      float *cmin = (float *)(ws + plan.cmin_off);
      HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)cmin, 0x7f800000, 32, st));
      hipLaunchKernelGGL(rn::delta_scale_min, dim3(rn::DSC_ROWS, 64), dim3(256), 0, st, a.delta, plan.S, cmin);
      const dim3 wf_grid(8 * ((slices + 7) / 8) * (rn::wf_tm(rn::WF_NW) == 256 ? rn::WJOBS_M256.tiles : rn::WJOBS.tiles));
      hipLaunchKernelGGL((rn::wgrad_f16s_kernel<rn::WF_NW>), wf_grid, dim3(64 * rn::WF_NW), rn::wf_lds(rn::WF_NW), st, w, slices, cmin);
    }
    else if (del16 && act16) hipLaunchKernelGGL((rn::wgrad_bf16x3_kernel<true, true>), wg_grid, dim3(256), rn::wb_lds(true, true), st, w, slices);
    else if (del16) hipLaunchKernelGGL((rn::wgrad_bf16x3_kernel<true, false>), wg_grid, dim3(256), rn::wb_lds(true, false), st, w, slices);
    else if (act16) hipLaunchKernelGGL((rn::wgrad_bf16x3_kernel<false, true>), wg_grid, dim3(256), rn::wb_lds(false, true), st, w, slices);
    else hipLaunchKernelGGL((rn::wgrad_bf16x3_kernel<false, false>), wg_grid, dim3(256), rn::wb_lds(false, false), st, w, slices);
  }
  else
    hipLaunchKernelGGL(rn::wgrad_kernel, dim3(rn::WJOBS.tiles, slices), dim3(256), (rn::WG_TM + rn::WG_TN) * rn::WG_LDK * 4, st, w);
  HIP_TRY(hipGetLastError());
  { int trc = timer_end(st, tslot); if (trc) return trc; }
  hipLaunchKernelGGL(rn::wgrad_reduce, dim3(1024), dim3(256), 0, st, w.part, slices, d_param_grads, (int)rn::NUM_PARAMS);
  if (gbasis) {
    /* the tail of the parameter blob: dW_ext[L] = DELTA(layer 0 | 5) x (IPE features of groups 1..)^T, same kernel on the
     * tail matrix behind ACT and its own job table, partials behind the seeds */
    rn::WgradArgs we = w;
    we.act = (const float *)((const char *)saved->d_activations + plan.act_ext_off);
    we.a_units = rn::ACT_EXT_UNITS;
    we.part = (float *)(ws + plan.part_ext_off);
    if (plan.pitch > plan.S)
      hipLaunchKernelGGL(rn::wgrad_zero_tail, dim3(256), dim3(256), 0, st, const_cast<float *>(we.act), rn::ACT_EXT_ROWS, rn::ACT_EXT_UNITS, plan.pitch, plan.S);
    /* fewer than 7 groups (icosahedron / 1, octahedron / 2): the forward wrote the rows of groups 1..G-1 only; the GEMM
     * contracts all 576, so the others must read as zeros -- their gradient is then exactly 0, as the header promises */
    if (cfg->ipe_groups - 1 < rn::EXT_GROUPS)
      hipLaunchKernelGGL(rn::wgrad_zero_units, dim3(2048), dim3(256), 0, st, const_cast<float *>(we.act), (cfg->ipe_groups - 1) * rn::IPE_DIM,
                         rn::ACT_EXT_ROWS, rn::ACT_EXT_UNITS, plan.pitch);
    hipLaunchKernelGGL((rn::wgrad_bf16x3_kernel<false, false, true>), dim3(8 * ((slices + 7) / 8) * rn::WJOBS_EXT.tiles), dim3(256), rn::wb_lds(false, false), st, we, slices);
    hipLaunchKernelGGL(rn::wgrad_reduce, dim3(256), dim3(256), 0, st, we.part, slices, d_param_grads + rn::NUM_PARAMS, (int)rn::EXT_PARAMS);
  }
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_sample_intervals(const float *d_t, const float *d_logits, int32_t R, int32_t M, int32_t N,
                             float s_min, float s_max, float *d_sdist, int32_t *d_bin_idx, void *stream) {
  if (!d_t || !d_logits || !d_sdist) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: null pointer%s");
  if (N <= 1) return fail(REFNERF_EINVAL, "num_samples must be > 1%s");
  if (M < 1 || M > 2048 || N > 2048 || R <= 0) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: size out of range%s");
  size_t lds = sizeof(float) * 4 * (size_t)((M + 1) + M + (M + 1) + N + (N + 1) + 8);
  if (lds > 160 * 1024) return fail(REFNERF_EINVAL, "refnerf_sample_intervals: M,N too large%s");
  LDS_ATTR_ONCE(lds_attr(rn::sample_intervals_kernel));
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
  hipLaunchKernelGGL(rn::ide_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_xyz, d_kappa_inv, n, d_ide);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_render_rays(const refnerf_level_cfg *cfg, int32_t R, const float *d_density, const float *d_tdist,
                        const float *d_directions, const float *d_far, const float *d_rgb, const float *d_diffuse,
                        const float *d_specular, const float *d_normals, const float *d_normals_pred,
                        const float *d_roughness, const float *d_tint, const refnerf_level_out *out, void *stream) {
  if (!cfg || !d_density || !d_tdist || !d_directions || !d_far || !out)
    return fail(REFNERF_EINVAL, "refnerf_render_rays: null pointer%s");
  const int N = cfg->n_samples;
  if (R <= 0 || N < 1 || N > 1024) return fail(REFNERF_EINVAL, "refnerf_render_rays: R must be positive and n_samples in [1,1024]%s");
  rn::RenderArgs a;
  a.cfg = *cfg;
  a.cfg.training = d_normals != nullptr;
  a.R = R;
  a.rpw = N <= 256 ? 4 : 1;
  a.density = d_density; a.tdist = d_tdist; a.dirs = d_directions; a.far = d_far;
  a.rgb = d_rgb; a.dif = d_diffuse; a.spc = d_specular; a.nrm = d_normals; a.npred = d_normals_pred;
  a.rough = d_roughness; a.tint = d_tint;
  a.out = *out;
  const size_t lds = sizeof(float) * ((size_t)a.rpw * (2 * (N + 1) + rn::NPS_TRAIN * N) + 4 * 22 * rn::WSUM_PITCH);
  LDS_ATTR_ONCE(lds_attr(rn::render_rays_kernel));
  hipLaunchKernelGGL(rn::render_rays_kernel, dim3((R + a.rpw - 1) / a.rpw), dim3(rn::NTHREADS), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_losses_forward(int32_t R, int32_t N, const float *d_r_rgb, const float *d_gt_rgb, const float *d_lossmult,
                           const float *d_weights, const float *d_orientation_normals, const float *d_normals,
                           const float *d_normals_pred, const float *d_viewdirs, float *d_terms, void *stream) {
  if (!d_r_rgb || !d_gt_rgb || !d_lossmult || !d_weights || !d_viewdirs || !d_terms || (d_normals && !d_normals_pred))
    return fail(REFNERF_EINVAL, "refnerf_losses_forward: null pointer%s");
  if (R <= 0 || N <= 0) return fail(REFNERF_EINVAL, "refnerf_losses_forward: R and N must be positive%s");
  rn::LossArgs a{};
  a.R = R; a.N = N; a.rgb = d_r_rgb; a.gt = d_gt_rgb; a.lossmult = d_lossmult; a.weights = d_weights;
  a.normals_o = d_orientation_normals; a.normals = d_normals; a.normals_pred = d_normals_pred; a.viewdirs = d_viewdirs;
  a.terms = d_terms;
  hipLaunchKernelGGL(rn::refnerf_losses_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_losses_backward(int32_t R, int32_t N, const float *d_r_rgb, const float *d_gt_rgb, const float *d_lossmult,
                            const float *d_weights, const float *d_orientation_normals, int32_t orientation_on_pred,
                            const float *d_normals, const float *d_normals_pred, const float *d_viewdirs,
                            float g_data, float g_orientation, float g_normal, const float *d_upstream,
                            float *d_g_r_rgb, float *d_g_weights, float *d_g_normals_pred, void *stream) {
  if (!d_r_rgb || !d_gt_rgb || !d_lossmult || !d_weights || !d_viewdirs || !d_g_r_rgb || !d_g_weights || !d_g_normals_pred ||
      (d_normals && !d_normals_pred))
    return fail(REFNERF_EINVAL, "refnerf_losses_backward: null pointer%s");
  if (R <= 0 || N < 3) return fail(REFNERF_EINVAL, "refnerf_losses_backward: R must be positive and N >= 3%s");
  rn::LossArgs a{};
  a.R = R; a.N = N; a.rgb = d_r_rgb; a.gt = d_gt_rgb; a.lossmult = d_lossmult; a.weights = d_weights;
  a.normals_o = d_orientation_normals; a.orient_on_pred = orientation_on_pred; a.normals = d_normals;
  a.normals_pred = d_normals_pred; a.viewdirs = d_viewdirs;
  a.g_data = g_data; a.g_orient = g_orientation; a.g_normal = g_normal; a.upstream = d_upstream;
  a.g_rgb = d_g_r_rgb; a.g_weights = d_g_weights; a.g_npred = d_g_normals_pred;
  const size_t n = (size_t)R * N;
  hipLaunchKernelGGL(rn::refnerf_losses_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return REFNERF_OK;
}

int refnerf_set_timing(int enable) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.mu);
  r.timing = enable != 0;
  r.events_used = 0;
  return REFNERF_OK;
}
int refnerf_get_timing_family(int family, double *total_ms, int64_t *launches) {
  Runtime &r = rt();
  std::lock_guard<std::mutex> lk(r.mu);
  double tot = 0.0;
  int64_t n = 0;
  for (size_t i = 0; i < r.events_used; ++i) {
    if (r.family[i] != family) continue;
    HIP_TRY(hipEventSynchronize(r.events[i].second));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, r.events[i].first, r.events[i].second));
    tot += ms;
    n += 1;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return REFNERF_OK;
}
int refnerf_get_timing(double *total_ms, int64_t *launches) {
  return refnerf_get_timing_family(REFNERF_TIMER_FORWARD, total_ms, launches);
}

}  /* extern "C" */
