/*
 * refnerf_pack_common.h -- canonical blob accessors shared by the weight packers of both translation units.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "refnerf_layout.h"

namespace rn {

/* W[row][k] of GEMM op `op` in canonical storage; k = canonical input column. */
__device__ inline float canon_w(const float *P, int op, int row, int k) {
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
__device__ inline float canon_b(const float *P, int op, int row) {
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

/* One plain 17 KB chunk of the 16-bit images (refnerf_layout.h): bias piece + 16 fragment pieces. */
template <typename E>
__device__ inline void fill_chunk_plain(const float *__restrict__ P, char *__restrict__ chunk, int op, int ob, int kind, bool first, int base) {
  /* bias piece: fp32 [h][16] (first chunk of the slice), rest of the KB zero */
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
    } else if (kind == BF_LDS8) {
      int kp = 16 * t + 8 * h + e;
      if (t < BF_IPE_REAL_KS) v = canon_w(P, op, row, base + kp /* LDS order = canonical IPE order */);
    } else {
      int kp = 16 * (t - 8) + 8 * h + e;
      /* dir k': [Re x36 | n.v | 0 0 0 | Im x36 | 0 0 0 0] */
      if (kp < IDE_TERMS) v = canon_w(P, op, row, base + BNECK + kp);
      else if (kp == IDE_TERMS) v = canon_w(P, op, row, base + BNECK + IDE_DIM);
      else if (kp >= 40 && kp < 40 + IDE_TERMS) v = canon_w(P, op, row, base + BNECK + IDE_TERMS + (kp - 40));
    }
    reinterpret_cast<E *>(chunk + 1024)[idx] = (E)v;
  }
}


/* Split-f16 image (REFNERF_PREC_F16X2): one chunk of the 16x16x32 spatial section (refnerf_layout.h; w = hi + lo,
 * hi = fl16(w), lo = fl16(w - hi)); the directional ops are plain f16 chunks */
__device__ inline void fill_chunk_sq(const float *__restrict__ P, char *__restrict__ chunk, int op, int ob, int kind, bool first, int base) {
  for (int e = threadIdx.x; e < 256; e += blockDim.x) {
    float v = 0.0f;
    if (e < 32 && first) {                       /* bias piece [T][b][4]: rows 16 T + 4 b + i of the slice */
      const int T = e >> 4, b = (e >> 2) & 3, i = e & 3;
      v = canon_b(P, op, ob * 32 + 16 * T + 4 * b + i);
    }
    reinterpret_cast<float *>(chunk)[e] = v;
  }
  for (int idx = threadIdx.x; idx < 16 * 512; idx += blockDim.x) {
    const int e = idx & 7, lane = (idx >> 3) & 63, pi = idx >> 9;
    const int bk = lane >> 4, r16 = lane & 15;
    int T, part, col;
    bool live = true;
    if (kind == SQ_A || kind == SQ_B || kind == SQ_X) {
      const int sl = pi >> 2, which = pi & 3;
      T = which & 1; part = which >> 1;
      if (kind == SQ_X) { live = sl < 3; col = base + 32 * sl + 8 * bk + e; }
      else { const int st = (kind == SQ_B ? 4 : 0) + sl; col = 32 * st + 16 * (e >> 2) + 4 * bk + (e & 3); }
    } else if (kind == SQ_BN) {
      const int st = pi >> 1;
      T = pi & 1; part = 0;
      col = 32 * st + 16 * (e >> 2) + 4 * bk + (e & 3);
    } else {                                     /* SQ_SC: tile T0 of the scalar block, [hi lo] per k-step */
      const int st = pi >> 1;
      T = 0; part = pi & 1;
      col = 32 * st + 16 * (e >> 2) + 4 * bk + (e & 3);
    }
    const float v = live ? canon_w(P, op, ob * 32 + 16 * T + r16, col) : 0.0f;
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16 *>(chunk + 1024)[idx] = part ? (_Float16)(v - (float)hi) : hi;
  }
}

}  // namespace rn
