/*
 * refnerf_level_bf16.h -- bf16-MFMA level kernel (v_mfma_f32_32x32x16_bf16,
 * fp32 accumulate).
 *
 * workgroup = 4 waves (1 wave/SIMD, whole register file), 256 samples per pass:
 * each wave owns 64 samples = two 32-sample MFMA column blocks; in the VALU
 * phases lane l IS sample l.  Per layer the wave computes D[256 out][64 samples]
 * = W x X with A = W fragments and B = activations:
 *   * activations stay in REGISTERS as packed bf16 B fragments (R0/R1 ping-pong,
 *     2 x 128 VGPR) -- the transposed formulation makes a layer's accumulator
 *     layout the next layer's B layout, the k permutation is folded into the
 *     weight image at pack time;
 *   * weights stream HBM/L2 -> LDS ONCE per workgroup with LDS-DMA
 *     (global_load_lds_dwordx4) in "slices" = (layer, 32-row output block),
 *     double-buffered, one __syncthreads per slice; all four waves read the
 *     same fragments from LDS (ds_read_b128, conflict-free lane-linear image);
 *   * encodings (IPE, IDE) are staged in LDS as bf16 [k/8][sample][8] so a B
 *     fragment is one ds_read_b128; the bottleneck stays in registers across
 *     the directional MLP (used by its layers 0 and 5).
 * Code size is kept small by running both MLP trunks through ONE rolled
 * "phase" loop whose body holds two generic layer instances (R0->R1, R1->R0).
 */
#pragma once
#include "refnerf_level_common.h"

namespace rn {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int BT = 256;                               /* samples per pass */
constexpr int WBUF_BYTES = BF_MAX_SLICE_KB * 1024;    /* one ring slot */
constexpr int BF_X_BYTES = (IPE_DIM / 8) * BT * 16;   /* 12 k-groups x 256 x 16 B = 48 KB */

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct BfCtx {
  const char *img;   /* packed bf16 image (global) */
  char *wbuf;        /* LDS ring: 2 x WBUF_BYTES */
  const char *xb;    /* LDS encodings */
  int cur;           /* ring slot holding the slice about to be consumed */
  int next_off;      /* byte offset of the next slice to prefetch */
  int lane, wave, h, n;
};

/* LDS-DMA of the next slice into the other ring slot: `pieces` x 1 KB, the four
 * waves take pieces round-robin (lane-linear 16 B per lane). */
__device__ __forceinline__ void issue_slice(BfCtx &c, int pieces) {
  char *dst = c.wbuf + (c.cur ^ 1) * WBUF_BYTES;
  const char *src = c.img + c.next_off + c.lane * 16;
  for (int p = c.wave; p < pieces; p += 4)
    __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024), (lptr_t)(dst + p * 1024), 16, 0, 0);
  c.next_off += pieces * 1024;
}

__device__ __forceinline__ v16f bias16(const char *w, int h) {
  const v4f *bp = reinterpret_cast<const v4f *>(w + h * 64);
  v4f b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
  return (v16f){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
}

/* A-fragment ring depth: ds_read of step k+AF is issued before the MFMAs of step
 * k (pinned with sched_barrier; hipcc otherwise serialises read->wait->mfma). */
constexpr int AF = 4;

__device__ __forceinline__ v8bf lds_frag(const char *p) { return *reinterpret_cast<const v8bf *>(p); }

/* One slice: acc[sb] = bias + W_slice x [in | bn | X_lds] for both sample blocks. */
template <bool HAS_REG>
__device__ __forceinline__ void bf_mma_slice(BfCtx &c, int next_pieces, bool has_bn, int ks_lds, int col0,
                                             const v8bf (&in)[2][16], const v8bf (&bn)[2][8], v16f (&acc)[2]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* my DMA pieces of this slice */
  __syncthreads();   /* the slice has landed for every wave; the other slot is free */
  issue_slice(c, next_pieces);
  const char *w = c.wbuf + c.cur * WBUF_BYTES;
  const char *wf = w + 1024 + c.lane * 16;
  v8bf a[AF];
#pragma unroll
  for (int d = 0; d < AF; ++d) a[d] = lds_frag(wf + d * 1024);
  acc[0] = bias16(w, c.h);
  acc[1] = acc[0];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (HAS_REG) {
#pragma unroll
    for (int k = 0; k < BF_REG_KS; ++k) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AF], in[0][k], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AF], in[1][k], acc[1], 0, 0, 0);
      a[k % AF] = lds_frag(wf + (k + AF) * 1024);    /* may run past the slice: same ring slot, unused */
      __builtin_amdgcn_sched_barrier(0);
    }
    wf += BF_REG_KS * 1024;
  }
  constexpr int P0 = HAS_REG ? (BF_REG_KS % AF) : 0;   /* ring phase (0: AF divides 16) */
  static_assert(P0 == 0 && BF_BN_KS % AF == 0, "ring phase");
  if (has_bn) {
#pragma unroll
    for (int k = 0; k < BF_BN_KS; ++k) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AF], bn[0][k], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AF], bn[1][k], acc[1], 0, 0, 0);
      a[k % AF] = lds_frag(wf + (k + AF) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    wf += BF_BN_KS * 1024;
  }
  const char *xp = c.xb + (c.h * BT + col0 + c.n) * 16;
#pragma unroll 1
  for (int k = 0; k < ks_lds; ++k) {
    v8bf x0 = lds_frag(xp + (2 * k) * BT * 16);
    v8bf x1 = lds_frag(xp + (2 * k) * BT * 16 + 32 * 16);
    v8bf af = a[0];
#pragma unroll
    for (int d = 0; d + 1 < AF; ++d) a[d] = a[d + 1];
    a[AF - 1] = lds_frag(wf + (k + AF) * 1024);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, x0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, x1, acc[1], 0, 0, 0);
  }
  c.cur ^= 1;
}

/* ReLU on the raw bits (max_i32(x, 0): negative floats are negative ints; no
 * canonicalising v_max pair) and v_cvt_pk_bf16_f32 pinned in place so hipcc
 * cannot keep the fp32 values alive until the next layer. */
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
template <bool RELU>
__device__ __forceinline__ void pack_acc(const v16f &a, v8bf &f0, v8bf &f1) {
  typedef unsigned v4uu __attribute__((ext_vector_type(4)));
  v4uu p0, p1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x0 = a[2 * e], x1 = a[2 * e + 1], y0 = a[8 + 2 * e], y1 = a[8 + 2 * e + 1];
    if (RELU) {
      x0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x0), 0));
      x1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x1), 0));
      y0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, y0), 0));
      y1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, y1), 0));
    }
    p0[e] = cvt_pk_bf16(x0, x1);
    p1[e] = cvt_pk_bf16(y0, y1);
  }
  f0 = __builtin_bit_cast(v8bf, p0);
  f1 = __builtin_bit_cast(v8bf, p1);
}

/* One 256-wide layer: 8 slices, ReLU, repack as next-layer B fragments. */
template <bool HAS_REG>
__device__ __forceinline__ void bf_layer(BfCtx &c, int pieces_self, int pieces_after, bool has_bn, int ks_lds,
                                         int col0, const v8bf (&in)[2][16], const v8bf (&bn)[2][8], v8bf (&out)[2][16]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    v16f acc[2];
    bf_mma_slice<HAS_REG>(c, ob < 7 ? pieces_self : pieces_after, has_bn, ks_lds, col0, in, bn, acc);
    pack_acc<true>(acc[0], out[0][2 * ob], out[0][2 * ob + 1]);
    pack_acc<true>(acc[1], out[1][2 * ob], out[1][2 * ob + 1]);
  }
}

__device__ __forceinline__ void st_bf16(char *p, float v) { *reinterpret_cast<__bf16 *>(p) = (__bf16)v; }

__global__ __launch_bounds__(NTHREADS) void level_fwd_bf16(const LevelArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rpw = A.rpw;
  const int ray0 = blockIdx.x * rpw;
  const int n_tot = rpw * N;

  char *WB = reinterpret_cast<char *>(smem);                 /* 2 x WBUF_BYTES          */
  char *Xb = WB + 2 * WBUF_BYTES;                            /* BF_X_BYTES: encodings   */
  float *HD = reinterpret_cast<float *>(Xb + BF_X_BYTES);    /* [HD_ROWS][BT]           */
  float *TD = HD + HD_ROWS * BT;                             /* [rpw][N+1]              */
  float *XP = TD + rpw * (N + 1);                            /* [rpw][N+1]              */
  float *PS = XP + rpw * (N + 1);                            /* [NPS][n_tot]            */

  BfCtx c;
  c.img = reinterpret_cast<const char *>(A.packed);
  c.wbuf = WB;
  c.xb = Xb;
  c.cur = 1;           /* so the prologue prefetch lands in slot 0 */
  c.next_off = 0;
  c.lane = lane; c.wave = wave; c.h = lane >> 5; c.n = lane & 31;

  issue_slice(c, BFPACKED.op[0].ks + 1);                    /* overlaps with the resampler */
  c.cur = 0;

  resample_phase(A, reinterpret_cast<float *>(Xb), TD, ray0, wave, lane);   /* P0 */
  __syncthreads();

  const int col0 = wave * 64;
  const int col = col0 + lane;
  v8bf R0[2][16], R1[2][16], bn[2][8];

  for (int pass0 = 0; pass0 < n_tot; pass0 += BT) {
    const int g = pass0 + col;
    const int rl = g / N, si = g - rl * N;
    const int ray = ray0 + rl;
    const bool valid = (g < n_tot) && (ray < A.R);
    const int rayc = valid ? ray : (A.R - 1);
    const bool last_pass = (pass0 + BT >= n_tot);
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = A.rays.d_viewdirs[(size_t)rayc * 3 + i];
    SampleHeads sh;
    int op = 0;

#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      if (phase == 0) {
        /* P1: conical frustum -> lifted Gaussian -> IPE in k' = 6j + 3c + b order */
        float o[3], d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] = A.rays.d_origins[(size_t)rayc * 3 + i]; d[i] = A.rays.d_directions[(size_t)rayc * 3 + i]; }
        float radius = A.rays.d_radii[rayc];
        const float *td = TD + (valid ? rl : 0) * (N + 1);
        float t0 = td[valid ? si : 0], t1 = td[valid ? si + 1 : 1];
        float lm[3], lv[3];
        cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
#pragma unroll 1
        for (int jq = 0; jq < 4; ++jq) {
          float f[24];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int cb = 0; cb < 6; ++cb) f[jj * 6 + cb] = ipe_feature(lm[cb % 3], lv[cb % 3], jq * 4 + jj, cb / 3);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            v8bf pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = (__bf16)f[q * 8 + e];
            *reinterpret_cast<v8bf *>(Xb + ((jq * 3 + q) * BT + col) * 16) = pk;
          }
        }
      } else {
        /* P4: head activations, reflection, IDE (k' = IDE index), n.v, zero pad */
        float gp[3], raw_dif[3], raw_tint[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          gp[i] = HD[(1 + i) * BT + col];
          raw_dif[i] = HD[(5 + i) * BT + col];
          raw_tint[i] = HD[(8 + i) * BT + col];
        }
        sample_heads(cfg, HD[0 * BT + col], gp, HD[4 * BT + col], raw_dif, raw_tint, v, sh);
        char *xs = Xb + col * 16;
#pragma unroll 1
        for (int part = 0; part < 2; ++part)
          ide_eval(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, part, [&](int q, float val) {
            int kq = part * IDE_TERMS + q;
            st_bf16(xs + (kq >> 3) * BT * 16 + (kq & 7) * 2, val);
          });
        v8bf tail;
        tail[0] = (__bf16)sh.dot;
#pragma unroll
        for (int e = 1; e < 8; ++e) tail[e] = (__bf16)0.0f;
        *reinterpret_cast<v8bf *>(xs + 9 * BT * 16) = tail;
      }
      wave_sync();

      /* layer 0 of the trunk: inputs from LDS (+ bottleneck registers for the dir MLP) */
      {
        const int self = BFPACKED.op[phase ? 9 : 0].ks + 1;
        bf_layer<false>(c, self, BF_REG_KS + 1, phase == 1, phase ? BF_DIR_KS : BF_IPE_KS, col0, R0, bn, R0);
        op += 1;
      }
      /* layers 1..7: A (R0->R1), B (R1->R0); the third A carries the skip input */
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const bool skip = (it == 2);
        const bool a_bn = skip && phase == 1;
        const int a_lds = skip ? (phase ? BF_DIR_KS : BF_IPE_KS) : 0;
        const int a_self = BF_REG_KS + (a_bn ? BF_BN_KS : 0) + a_lds + 1;
        /* what follows layer A: B (it<3) or heads / rgb (16 reg steps either way) */
        bf_layer<true>(c, a_self, BF_REG_KS + 1, a_bn, a_lds, col0, R0, bn, R1);
        op += 1;
        if (it < 3) {
          /* what follows layer B: the next A, which is the skip layer when it == 1 */
          const int nxt = BF_REG_KS + 1 + ((it == 1) ? ((phase ? BF_BN_KS + BF_DIR_KS : BF_IPE_KS)) : 0);
          bf_layer<true>(c, BF_REG_KS + 1, nxt, false, 0, col0, R1, bn, R0);
          op += 1;
        }
      }
      if (phase == 0) {
        /* P3: heads: 4 bottleneck blocks stay in registers, scalar block -> LDS HD */
#pragma unroll
        for (int ob = 0; ob < 5; ++ob) {
          v16f acc[2];
          const int nxt = (ob < 4) ? BF_REG_KS + 1 : BFPACKED.op[9].ks + 1;
          bf_mma_slice<true>(c, nxt, false, 0, col0, R1, bn, acc);
          if (ob < 4) {
            pack_acc<false>(acc[0], bn[0][2 * ob], bn[0][2 * ob + 1]);
            pack_acc<false>(acc[1], bn[1][2 * ob], bn[1][2 * ob + 1]);
          } else {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * c.h;
                if (row < HD_ROWS) HD[row * BT + col0 + sb * 32 + c.n] = acc[sb][r];
              }
          }
        }
        op += 1;
        wave_sync();
      } else {
        /* rgb: one slice; the prefetch behind it wraps to the first slice of the next pass */
        v16f acc[2];
        c.next_off = 0;
        bf_mma_slice<true>(c, last_pass ? 0 : BFPACKED.op[0].ks + 1, false, 0, col0, R1, bn, acc);
        float raw_rgb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float a0 = __shfl(acc[0][i], c.n, 64), a1 = __shfl(acc[1][i], c.n, 64);
          raw_rgb[i] = (lane < 32) ? a0 : a1;
        }
        if (valid) colour_store(A, sh, raw_rgb, PS, n_tot, g, (size_t)ray * N + si);   /* P6 */
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  composite_phase(A, TD, XP, PS, n_tot, ray0, wave, lane);   /* P7 */
}

/* ---------------- bf16 weight image ---------------- */
__device__ __forceinline__ int ipe_col_of_kprime(int kp) {        /* k' = 6j + 3c + b */
  int j = kp / 6, cb = kp % 6;
  return (cb / 3) * 48 + j * 3 + (cb % 3);
}

}  // namespace rn
