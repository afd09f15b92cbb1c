/*
 * refnerf_level_bf16.h -- bf16-MFMA level kernel (v_mfma_f32_32x32x16_bf16,
 * fp32 accumulate).
 *
 * workgroup = 8 waves x 32 samples = 256 samples per pass, ONE workgroup per CU,
 * two waves per SIMD, <= 256 registers per lane (everything in arch VGPRs: hipcc
 * copies MFMA A/B operands out of AGPRs instead of using them in place, which
 * sank a 64-samples/wave variant).  256 samples per weight byte keep the
 * weight stream at ~17 B/clk/CU at full MFMA rate (128 samples x 2 workgroups
 * per CU measured LDS-DMA-bound at 40 % MFMA busy).
 * Per layer a wave computes D[256 out][32 samples] = W x X, A = W fragments,
 * B = activations:
 *   * activations stay in REGISTERS as packed bf16 B fragments (R0/R1 ping-pong,
 *     2 x 64 VGPR): the transposed formulation makes a layer's accumulator
 *     layout the next layer's B layout (k permutation folded into the weight
 *     image at pack time);
 *   * weights stream HBM/L2 -> LDS once per workgroup by LDS-DMA
 *     (global_load_lds_dwordx4) in uniform 17 KB chunks through a 3-slot ring;
 *     ONE barrier per chunk, placed MID-chunk: it certifies chunk c+1 (DMA'd a
 *     whole chunk-time earlier) and frees the slot of chunk c-1 for the DMA of
 *     c+2, so the A-fragment ring (ds_read_b128, 4 steps ahead) runs straight
 *     across chunk/slice/layer boundaries; while one wave of a SIMD repacks
 *     its accumulators the other one feeds the matrix pipe;
 *   * encodings (IPE, IDE) are staged in LDS as bf16 [k/8][sample][8] so a B
 *     fragment is one ds_read_b128; the bottleneck stays in registers across
 *     the directional MLP (used by its layers 0 and 5).
 * Both MLP trunks run through ONE rolled "phase" loop whose body holds two
 * generic layer instances (R0->R1, R1->R0) to bound code size.
 */
#pragma once
#include <type_traits>

#include "refnerf_level_common.h"

namespace rn {

typedef short v2s __attribute__((ext_vector_type(2)));

/* Element type of the MFMA operands: the kernel is a template over it.
 *   MmBf16: v_mfma_f32_32x32x16_bf16 (8 significand bits, fp32 range)            -- REFNERF_PREC_BF16
 *   MmF16 : v_mfma_f32_32x32x16_f16  (IEEE half: 11 significand bits, |x| <= 65504) -- REFNERF_PREC_F16
 * Same MFMA rate, same packed layouts (two 16-bit values per dword), fp32 accumulation in both.  f16 is 8-10x closer
 * to the fp32 parity mode (measured); it needs the hidden activations to stay below 65504 (they are O(1)..O(100)
 * in a NeRF MLP; the head outputs -- densities, colours -- never pass through the 16-bit type). */
struct MmBf16 {
  typedef __bf16 t;
  typedef t v8 __attribute__((ext_vector_type(8)));
  typedef t v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ v16f mfma(v8 a, v8 b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct MmF16 {
  typedef _Float16 t;
  typedef t v8 __attribute__((ext_vector_type(8)));
  typedef t v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ v16f mfma(v8 a, v8 b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <typename MM>
__device__ __forceinline__ unsigned cvt_pk_mm(float lo, float hi) {
  typename MM::v2 r = __builtin_convertvector((v2f){lo, hi}, typename MM::v2);
  return __builtin_bit_cast(unsigned, r);
}
#ifdef REFNERF_EVAL_EXACT_ENC
constexpr bool ENC_FAST = false;     /* experiment: libm-accurate encodings / head activations in this kernel */
#else
constexpr bool ENC_FAST = true;
#endif

constexpr int BT = 256;                               /* samples per pass: 8 waves x 32 */
constexpr int BF_NW = 8;                              /* waves per workgroup */
constexpr int BF_NTHREADS = 64 * BF_NW;
constexpr int BF_RING_BYTES = 3 * BF_CHUNK_BYTES;     /* 51 KB */
constexpr int BF_X_BYTES = (IPE_DIM / 8) * BT * 16;   /* 12 k-groups x 256 x 16 B = 48 KB */
#ifndef REFNERF_BF_AF
#define REFNERF_BF_AF 2
#endif
constexpr int AF = REFNERF_BF_AF;                     /* A-fragment ring depth (k-steps ahead) */
static_assert(AF == 1 || AF == 2 || AF == 4, "the ring index k % AF must stay in phase across 8- and 16-step chunks, and the ring may only run into "
                                             "the next chunk after that chunk's rendezvous (8-step chunks: k >= 4)");
/* measured in the full kernel (C2, round 2): AF = 2 runs 1.2 % faster than 4 (4 fewer fragment registers: 36 instead of
 * 84-100 bytes of scratch per lane); in the isolated MLP loop (scripts/mb) the deeper ring is the faster one */

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct Pipe {
  const char *src;   /* this lane's DMA source inside the image: advances one chunk per rendezvous */
  const char *src_end;   /* end of one pass worth of chunks (wrap point) for this lane */
  char *wbuf;        /* LDS ring base (3 slots) */
  const char *xp;    /* LDS encodings, pre-offset to this lane's B fragment (sample n, half h) */
  int cur_off, nxt_off, fil_off;   /* ring slots: being consumed / landed next / free */
  int dma_left;      /* chunks still to be DMA'd by this workgroup */
  int lane, wave, h;
  long long t_vm, t_bar;   /* debug (REFNERF_PROF): cycles spent in the DMA wait / in the barrier */
};

/* LDS-DMA of one 17 KB chunk into the ring slot at `slot_off`.  Wave w moves the
 * adjacent pieces 3w..3w+2 (waves 0-4; wave 5 moves 15,16): one address and the
 * instruction's immediate offset cover both the global and the LDS side.
 * (Measured alternatives: all pieces issued by the prioritised waves 4-7, or
 * 2 pieces per wave with 16 KB chunks -- both slower in the full kernel.) */
__device__ __forceinline__ void issue_chunk(Pipe &p, int slot_off) {
  if (p.dma_left > 0) {
    if (p.wave < 6) {
      lptr_t dst = (lptr_t)(p.wbuf + slot_off + p.wave * 3072);
      __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 1024, 0);
      if (p.wave < 5) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 2048, 0);
    }
    p.src += BF_CHUNK_BYTES;
    if (p.src == p.src_end) p.src -= (size_t)BFPACKED.chunks_per_pass * BF_CHUNK_BYTES;
    p.dma_left -= 1;
  }
}

template <typename MM>
__device__ __forceinline__ typename MM::v8 lds_frag(const char *q) { return *reinterpret_cast<const typename MM::v8 *>(q); }

__device__ __forceinline__ v16f bias16(const char *w, int h) {
  const v4f *bp = reinterpret_cast<const v4f *>(w + h * 64);
  v4f b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
  return (v16f){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3],
                b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
}

template <typename MM, bool RELU>
__device__ __forceinline__ unsigned pack_pair(float lo, float hi) {
  unsigned w = cvt_pk_mm<MM>(lo, hi);
  if (RELU) {
    v2s s = __builtin_bit_cast(v2s, w);
    s = __builtin_elementwise_max(s, (v2s){0, 0});
    w = __builtin_bit_cast(unsigned, s);
  }
  return w;
}

/* acc (one 32x32 fp32 tile) -> two packed bf16 B fragments of the next layer */
template <typename MM, bool RELU>
__device__ __forceinline__ void pack_acc(const v16f &a, v4uu &f0, v4uu &f1) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f0[e] = pack_pair<MM, RELU>(a[2 * e], a[2 * e + 1]);
    f1[e] = pack_pair<MM, RELU>(a[8 + 2 * e], a[8 + 2 * e + 1]);
  }
}

/* B fragment (LDS encodings) of LDS k-step kl; REAL steps hold data, the
 * zero-weight pad steps re-read early groups (any finite value). */
template <typename MM, int REAL>
__device__ __forceinline__ typename MM::v8 lds_b(const Pipe &p, int kl) {
  const int kk = (kl < REAL) ? kl : kl - REAL;
  return lds_frag<MM>(p.xp + (2 * kk) * BT * 16);
}

/* One chunk.  KIND: BF_REG (16 steps over `in`), BF_LDS8 (8 steps over LDS,
 * REAL_L real), BF_BNLDS (8 over `bn` + 8 over LDS, REAL_L real).
 * FIRST: the chunk opens a slice (accumulator starts from the bias piece).
 * `a` is the A-fragment ring; on entry it holds fragments 0..AF-1 of this chunk,
 * on exit those of the next one. */
template <typename MM, int KIND, int REAL_L, bool FIRST>
__device__ __forceinline__ void bf_chunk(Pipe &p, typename MM::v8 (&a)[AF], const v4uu (&in)[16], const v4uu (&bn)[8], v16f &acc) {
  typedef typename MM::v8 v8mm;
  constexpr int KS = (KIND == BF_LDS8) ? 8 : 16;
  constexpr int L0 = (KIND == BF_BNLDS) ? 8 : 0;      /* first LDS step (for the LDS kinds) */
  const char *w = p.wbuf + p.cur_off;
  const char *cur = w + 1024 + p.lane * 16;
  const char *nxt = p.wbuf + p.nxt_off + 1024 + p.lane * 16;
  v8mm xr[2];
  if (KIND == BF_LDS8) { xr[0] = lds_b<MM, REAL_L>(p, 0); xr[1] = lds_b<MM, REAL_L>(p, 1); }
  if (FIRST) acc = bias16(w, p.h);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    v8mm b;
    const bool lds_step = (KIND == BF_LDS8) || (KIND == BF_BNLDS && k >= 8);
    if (lds_step) b = xr[(k - L0) & 1];
    else if (KIND == BF_REG) b = __builtin_bit_cast(v8mm, in[k]);
    else b = __builtin_bit_cast(v8mm, bn[k & 7]);
    acc = MM::mfma(a[k % AF], b, acc);
    /* A ring: fragment k+AF of this chunk, or the head of the next chunk (landed: k >= KS/2) */
    a[k % AF] = (k + AF < KS) ? lds_frag<MM>(cur + (k + AF) * 1024) : lds_frag<MM>(nxt + (k + AF - KS) * 1024);
    if (KIND != BF_REG) {
      const int kl2 = k + 2 - L0;                       /* LDS step to fetch now */
      if (kl2 >= 0 && kl2 < 8 && !(KIND == BF_LDS8 && kl2 < 2)) xr[kl2 & 1] = lds_b<MM, REAL_L>(p, kl2);
    }
#ifndef REFNERF_BF_RDV
#define REFNERF_BF_RDV (KS / 2 - 1)
#endif
    if (k == REFNERF_BF_RDV) {
      /* mid-chunk rendezvous: chunk c+1 is complete for every wave, chunk c-1's slot is free */
#ifdef REFNERF_PROF_WAITS
      long long t0 = (long long)__builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      long long t1 = (long long)__builtin_readcyclecounter();
      __syncthreads();
      long long t2 = (long long)__builtin_readcyclecounter();
      p.t_vm += t1 - t0;
      p.t_bar += t2 - t1;
#else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#endif
      issue_chunk(p, p.fil_off);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const int t = p.cur_off;
  p.cur_off = p.nxt_off;
  p.nxt_off = p.fil_off;
  p.fil_off = t;
}

/* A wave whose 32 samples of this pass are all past the end (the partly filled last pass of a workgroup: N = 192 puts 384
 * samples into 2 x 256) takes no part in the MLP, but keeps the weight stream going: it joins every rendezvous of the pass
 * and issues its DMA pieces.  The matrix pipe of its SIMD then belongs to the one active wave -- a half-filled pass costs
 * about two thirds of a full one instead of all of it. */
__device__ __forceinline__ void idle_pass(Pipe &p) {
#pragma unroll 1
  for (int c = 0; c < BFPACKED.chunks_per_pass; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue_chunk(p, p.fil_off);
    const int t = p.cur_off;
    p.cur_off = p.nxt_off;
    p.nxt_off = p.fil_off;
    p.fil_off = t;
  }
}

/* One slice (32 output rows): first chunk of kind KIND0 plus, for the skip
 * layers, a run-time selected second chunk (1: IPE from LDS, 2: bottleneck +
 * dir encodings). */
template <typename MM, int KIND0, int REAL0>
__device__ __forceinline__ void bf_slice(Pipe &p, typename MM::v8 (&a)[AF], int second, const v4uu (&in)[16], const v4uu (&bn)[8], v16f &acc) {
  bf_chunk<MM, KIND0, REAL0, true>(p, a, in, bn, acc);
  if constexpr (KIND0 == BF_REG) {
    if (second == 1) bf_chunk<MM, BF_LDS8, BF_IPE_REAL_KS, false>(p, a, in, bn, acc);
    else if (second == 2) bf_chunk<MM, BF_BNLDS, BF_DIR_REAL_KS, false>(p, a, in, bn, acc);
  }
}

/* One 256-wide layer: 8 slices, ReLU, repack as next-layer B fragments.
 * The slice loop is ROLLED (one slice body per layer instance instead of eight:
 * the fully unrolled kernel was 80 KB of straight-line code cycling through a
 * 64 KB instruction cache, i.e. fetch-bound).  A rolled loop cannot index the
 * destination registers dynamically, so `out` works as a shift register: the
 * new fragments enter at [12..15] and everything moves down four places per
 * slice pair; after the 8 slices fragment pair ob sits at [2ob],[2ob+1]. */
template <typename MM, int KIND0, int REAL0>
__device__ __forceinline__ void bf_layer(Pipe &p, typename MM::v8 (&a)[AF], int second, const v4uu (&in)[16], const v4uu (&bn)[8], v4uu (&out)[16]) {
  /* fully unrolled: slice ob packs straight into out[2ob], out[2ob+1] (no register-queue moves) */
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    v16f acc;
    bf_slice<MM, KIND0, REAL0>(p, a, second, in, bn, acc);
    pack_acc<MM, true>(acc, out[2 * ob], out[2 * ob + 1]);
    __builtin_amdgcn_sched_barrier(0);   /* pack now: do not keep the fp32 tile alive */
  }
}

/* RINGPS: the workgroup owns more samples than per-sample records fit the LDS (rays_per_wg * N > 640: e.g. 4 rays x 192 =
 * three FULL passes instead of 2 rays = one and a half): the records live in a ring of BF_PS_RING rows and every ray is
 * composited right behind the pass that brings its last sample (N <= 256: a ray and the pass in flight fit the ring). */
constexpr int BF_PS_RING = 512;
template <typename MM, bool RINGPS = false>
__device__ __forceinline__ void level_fwd_mm(const LevelArgs &A) {
  constexpr int PSM = RINGPS ? BF_PS_RING - 1 : 0;
  typedef typename MM::v8 v8mm;
  typedef typename MM::t mm_t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rpw = A.rpw;
  const int ray0 = blockIdx.x * rpw;
  const int n_tot = rpw * N;
  const int n_pass = (n_tot + BT - 1) / BT;

  char *WB = reinterpret_cast<char *>(smem);                 /* 3 x 17 KB chunk ring     */
  char *Xb = WB + BF_RING_BYTES;                             /* BF_X_BYTES: encodings    */
  float *HD = reinterpret_cast<float *>(Xb + BF_X_BYTES);    /* [HD_ROWS][BT]            */
  float *TD = HD + HD_ROWS * BT;                             /* [rpw][N+1]               */
  float *XP = TD + rpw * (N + 1);                            /* [rpw][N+1]               */
  float *PS = XP + rpw * (N + 1);                            /* [n_tot][NPS_EVAL] (RINGPS: [BF_PS_RING][NPS_EVAL]) */
  float *PX = PS + (RINGPS ? BF_PS_RING : n_tot) * NPS_EVAL; /* [BT][3] grad_pred of the pass */
  float *NRM = PX + 3 * BT;                                  /* [8] |direction| per ray  */
  const float *RY = NRM + 8;                                 /* [rpw][12] o, d, viewdir, radius per ray */

  const int h = lane >> 5, n = lane & 31;
  const int col = wave * 32 + n;                             /* this lane's sample column */

  Pipe p;
  p.src = reinterpret_cast<const char *>(A.packed) + wave * 3072 + lane * 16;
  p.src_end = p.src + (size_t)BFPACKED.chunks_per_pass * BF_CHUNK_BYTES;
  p.wbuf = WB;
  p.xp = Xb + (h * BT + col) * 16;
  p.cur_off = 0; p.nxt_off = BF_CHUNK_BYTES; p.fil_off = 2 * BF_CHUNK_BYTES;
  p.dma_left = n_pass * BFPACKED.chunks_per_pass;
  p.lane = lane; p.wave = wave; p.h = h;
  p.t_vm = 0; p.t_bar = 0;
  RN_STAMP(A, 0);
  issue_chunk(p, p.cur_off);                                 /* overlaps with the resampler */
  issue_chunk(p, p.nxt_off);

  resample_phase<BF_NW, false>(A, reinterpret_cast<float *>(Xb), TD, NRM, ray0, wave, lane);   /* P0 */
  RN_STAMP(A, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           /* chunks 0 and 1 have landed */
  RN_STAMP(A, 2);

  /* static priority for the younger wave of each SIMD (waves 4-7): age-based
   * arbitration otherwise lets waves 0-3 run ahead and idle at every rendezvous */
#ifndef REFNERF_BF_NOPRIO
  if (wave >= BF_NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  v4uu R0[16], R1[16], bn[8];
  v8mm ar[AF];
#pragma unroll
  for (int d = 0; d < AF; ++d) ar[d] = lds_frag<MM>(WB + 1024 + lane * 16 + d * 1024);

  for (int pass0 = 0; pass0 < n_tot; pass0 += BT) {
    /* opaque copies: keep hipcc from hoisting ~100 registers of per-lane address
     * arithmetic out of the pass loop (it then spills them to scratch) */
    int lane_v = lane, col_v = col;
    asm volatile("" : "+v"(lane_v), "+v"(col_v));
    const int g = pass0 + col_v;
    const int rl = g / N, si = g - rl * N;
    const int ray = ray0 + rl;
    const bool valid = (g < n_tot) && (ray < A.R);
    const int rayc = valid ? ray : (A.R - 1);
    /* RINGPS: the rays whose last sample this pass brings are composited behind it (every wave meets the barrier) */
    auto pass_epilogue = [&]() {
      if constexpr (RINGPS) {
        __syncthreads();
        const int end = (pass0 + BT < n_tot) ? pass0 + BT : n_tot;
        composite_phase<BF_NW, true, NPS_EVAL, PSM>(A, TD, XP, PS, n_tot, ray0, wave, lane, nullptr, NRM, pass0 / N, end / N);
      }
    };
    {
      /* validity is monotone in the sample index: the wave is idle iff its first sample is past the end */
      const int g0 = pass0 + wave * 32;
      if (g0 >= n_tot || ray0 + g0 / N >= A.R) { idle_pass(p); pass_epilogue(); continue; }
    }
    /* head scalars of this sample live in LDS (HD); P4 and P6 both rebuild the
     * activations from them instead of keeping ~20 VGPRs alive across the dir MLP */
    auto load_heads = [&](SampleHeads &sh) {
      float v[3], gp[3], raw_dif[3], raw_tint[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v[i] = RY[(valid ? rl : 0) * 12 + 6 + i];
        gp[i] = HD[(1 + i) * BT + col];
        raw_dif[i] = HD[(5 + i) * BT + col];
        raw_tint[i] = HD[(8 + i) * BT + col];
      }
      sample_heads<ENC_FAST>(cfg, HD[0 * BT + col], gp, HD[4 * BT + col], raw_dif, raw_tint, v, sh);
    };

#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      /* tell the register allocator that the activation registers are dead across the
       * VALU phases (it cannot see through the rolled phase loop and would spill them) */
#pragma unroll
      for (int e = 0; e < 16; ++e) { R0[e] = (v4uu){0, 0, 0, 0}; R1[e] = (v4uu){0, 0, 0, 0}; }
      if (phase == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bn[e] = (v4uu){0, 0, 0, 0};
      }
      char *xs = Xb + col * 16;
      if (phase == 0) {
        /* P1: conical frustum -> lifted Gaussian -> IPE (half h computes block h: sin / cos) */
        float o[3], d[3];
        const float *ry = RY + (valid ? rl : 0) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] = ry[i]; d[i] = ry[3 + i]; }
        float radius = ry[9];
        const float *td = TD + (valid ? rl : 0) * (N + 1);
        float t0 = td[valid ? si : 0], t1 = td[valid ? si + 1 : 1];
        float lm[3], lv[3];
        cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
        if (cfg.disable_integration) { lv[0] = 0.0f; lv[1] = 0.0f; lv[2] = 0.0f; }        /* models.py:228-231 */
        /* k' = canonical IPE index: half h owns block h (sin / cos) = k' 48h .. 48h+47 = 6 k-groups;
         * rolled over two halves of 24 features (8 degrees x 3 axes): the (axis, degree) pattern repeats */
        RN_STAMP(A, 17);
#pragma unroll 1
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            v8mm pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int kk = q * 8 + e;                  /* 3 * (j - 8qq) + b */
              pk[e] = (mm_t)ipe_feature<ENC_FAST>(lm[kk % 3], lv[kk % 3], 8 * qq + kk / 3, h);
            }
            *reinterpret_cast<v8mm *>(xs + (6 * h + 3 * qq + q) * BT * 16) = pk;
          }
        }
        RN_STAMP(A, 18);
      } else {
        /* P4: head activations, reflection, IDE (k' = IDE index; half 0 real, half 1 imaginary) */
        SampleHeads sh;
        load_heads(sh);
        /* dir k' layout: [Re x36 | n.v | 0 0 0 | Im x36 | 0 0 0 0] = 2 x 5 k-groups; half h packs its 5 */
        float ide[40];
#pragma unroll
        for (int q = 36; q < 40; ++q) ide[q] = 0.0f;
        if (cfg.dir_enc == REFNERF_DIRENC_POSENC) posenc_eval<ENC_FAST>(sh.refd[0], sh.refd[1], sh.refd[2], h, [&](int q, float val) { ide[q] = val; });
        else ide_eval<ENC_FAST>(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, h, [&](int q, float val) { ide[q] = val; });
        if (h == 0) ide[36] = sh.dot;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          v8mm pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) pk[e] = (mm_t)ide[q * 8 + e];
          *reinterpret_cast<v8mm *>(xs + (5 * h + q) * BT * 16) = pk;
        }
      }
      wave_sync();
      RN_STAMP(A, 3 + phase * 8);

      /* layer 0 of the trunk: inputs from LDS (+ bottleneck registers for the dir MLP) -> R0 */
      if (phase == 0) bf_layer<MM, BF_LDS8, BF_IPE_REAL_KS>(p, ar, 0, R0, bn, R0);
      else bf_layer<MM, BF_BNLDS, BF_DIR_REAL_KS>(p, ar, 0, R0, bn, R0);
      RN_STAMP(A, 4 + phase * 8);
      /* layers 1..7: A (R0->R1), B (R1->R0); the third A carries the skip input */
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const int second = (it == 2) ? (phase ? 2 : 1) : 0;
        bf_layer<MM, BF_REG, 0>(p, ar, second, R0, bn, R1);
        if (it < 3) bf_layer<MM, BF_REG, 0>(p, ar, 0, R1, bn, R0);
      }
      RN_STAMP(A, 5 + phase * 8);
      if (phase == 0) {
        /* P3: heads: 4 bottleneck blocks stay in registers, the scalar block goes to LDS HD */
#pragma unroll
        for (int ob = 0; ob < 5; ++ob) {
          v16f acc;
          bf_slice<MM, BF_REG, 0>(p, ar, 0, R1, bn, acc);
          if (ob < 4) pack_acc<MM, false>(acc, bn[2 * ob], bn[2 * ob + 1]);
          else {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              int row = (rr & 3) + 8 * (rr >> 2) + 4 * h;
              if (row < HD_ROWS) HD[row * BT + col] = acc[rr];
            }
          }
        }
        wave_sync();
        RN_STAMP(A, 6);
      } else {
        /* rgb: one slice */
        v16f acc;
        bf_slice<MM, BF_REG, 0>(p, ar, 0, R1, bn, acc);
        float raw_rgb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(acc[i], n, 64);
        /* opaque copies taken HERE: the per-lane output addresses of P6 must not be computed at
         * the top of the pass and carried (spilled) across both MLP phases */
        int lane_w = lane, g_w = g, pass_w = pass0;
        asm volatile("" : "+v"(lane_w), "+v"(g_w), "+s"(pass_w));
        if (valid && h == 0) {                                                            /* P6 */
          SampleHeads sh;
          load_heads(sh);
          colour_store<ENC_FAST, NPS_EVAL, PSM>(A, sh, raw_rgb, PS, PX, n_tot, g_w, col);
        }
        wave_sync();
        history_flush<NPS_EVAL, PSM>(A, PS, PX, n_tot, pass_w + wave * 32, wave * 32, (size_t)ray0 * N + pass_w + wave * 32, lane_w);
        RN_STAMP(A, 14);
      }
    }
    __builtin_amdgcn_wave_barrier();
    pass_epilogue();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RN_STAMP(A, 15);
#ifdef REFNERF_PROF_WAITS
  if (A.prof && blockIdx.x == 0 && lane == 0) { A.prof[wave * 32 + 20] = p.t_vm; A.prof[wave * 32 + 21] = p.t_bar; }
#endif

  if constexpr (!RINGPS) composite_phase<BF_NW, true, NPS_EVAL>(A, TD, XP, PS, n_tot, ray0, wave, lane, reinterpret_cast<float *>(WB), NRM);   /* P7 */
  RN_STAMP(A, 16);
}

__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_bf16(const LevelArgs A) { level_fwd_mm<MmBf16>(A); }
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16(const LevelArgs A) { level_fwd_mm<MmF16>(A); }
/* the same with the per-sample records in a ring (rays_per_wg * N > 640) */
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_bf16_ring(const LevelArgs A) { level_fwd_mm<MmBf16, true>(A); }
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16_ring(const LevelArgs A) { level_fwd_mm<MmF16, true>(A); }

/* ---------------- bf16 weight image ---------------- */
__device__ __forceinline__ int ipe_col_of_kprime(int kp) { return kp; }   /* LDS order = canonical IPE order */

}  // namespace rn
