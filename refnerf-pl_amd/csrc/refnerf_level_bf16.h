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
/* REFNERF_RING_SLOTS = 4 (round 6): a fourth ring slot = TWO chunk-times between the issue of a chunk's LDS-DMA and the rendezvous
 * that certifies it (the 3-slot ring gives one) */
#ifndef REFNERF_RING_SLOTS
#define REFNERF_RING_SLOTS 3
#endif
constexpr int BF_RING_BYTES = REFNERF_RING_SLOTS * BF_CHUNK_BYTES;     /* 51 KB (68 KB with four slots) */
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
  const char *xps;   /* split mode: this lane's B fragment in the IPE planes of a run (hi plane; the lo plane (BT / 2) * 16 bytes behind) */
  int seq;           /* split mode: chunks issued so far in this pass (the spatial section is streamed twice) */
  int cur_off, nxt_off, fil_off;   /* ring slots: being consumed / landed next / free */
  int nx2_off;                     /* four-slot ring: the chunk behind `nxt` (in flight or landed) */
  int dma_left;      /* chunks still to be DMA'd by this workgroup */
  int lane, wave, h;
  long long t_vm, t_bar;   /* debug (REFNERF_PROF): cycles spent in the DMA wait / in the barrier */
};

/* LDS-DMA of one 17 KB chunk into the ring slot at `slot_off`.  Wave w moves the
 * adjacent pieces 3w..3w+2 (waves 0-4; wave 5 moves 15,16): one address and the
 * instruction's immediate offset cover both the global and the LDS side.
 * (Measured alternatives: all pieces issued by the prioritised waves 4-7, or
 * 2 pieces per wave with 16 KB chunks -- both slower in the full kernel.) */
/* REFNERF_BF_SPREAD: the (up to) three pieces a wave moves per chunk are issued one at a time, two k-steps apart, instead of
 * back to back right behind the rendezvous (an LDS-DMA piece costs its wave 100-185 issue cycles inside a burst, ~60 among
 * MFMAs: MI355X_MICROARCH.md); `piece` = 0, 1, 2, or -1 for all three.  Measured (round 4, C2, same box): f16x2 4.865 -> 4.839 ms
 * per step, bf16 2.105 -> 2.086, f16 2.175 -> 2.153, results bit-identical: on. */
#ifndef REFNERF_BF_SPREAD
#define REFNERF_BF_SPREAD 1
#endif
#ifndef REFNERF_DMA_AUX
#define REFNERF_DMA_AUX 0   /* cache policy bits of the weight stream's LDS-DMA */
#endif
/* the rendezvous of a chunk: this wave's DMA pieces have landed, then every wave's.  REFNERF_BARE_BARRIER: s_barrier alone behind
 * the wait -- __syncthreads() is a workgroup fence + barrier, and the fence is `s_waitcnt vmcnt(0) lgkmcnt(0)`: it also drains
 * the A-fragment reads that were issued ahead for the MFMAs BEHIND the rendezvous */
#ifndef REFNERF_BARE_BARRIER
#define REFNERF_BARE_BARRIER 0
#endif
#if REFNERF_RING_SLOTS == 4
/* four slots: chunk c + 1 must have landed, chunk c + 2 (<= 3 pieces per wave, wave 5: 2) may still fly: vmcnt(2) certifies c + 1 for
 * every wave (loads return in order).  No __syncthreads(): its fence is vmcnt(0); wavefront-scope fences keep the allocator sane. */
#define RN_RENDEZVOUS() do { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#elif REFNERF_BARE_BARRIER
/* (round 6) the wavefront-scope fences emit no instruction, but without them the register allocator spills 129-154 VGPRs in
 * these kernels: round 5's "bare barrier is 11 % slower" was that spill.  Measured with the fences (0 spills): f16x2 3.86 ->
 * 3.86 ms per C2 step, bf16 2.077 -> 2.067: the lgkmcnt(0) drain of __syncthreads()' fence costs nothing here; it stays. */
#define RN_RENDEZVOUS() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define RN_RENDEZVOUS() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#endif
__device__ __forceinline__ void ring_rotate(Pipe &p) {
  const int t = p.cur_off;
  p.cur_off = p.nxt_off;
#if REFNERF_RING_SLOTS == 4
  p.nxt_off = p.nx2_off;
  p.nx2_off = p.fil_off;
#else
  p.nxt_off = p.fil_off;
#endif
  p.fil_off = t;
}
template <bool SPLIT = false>
__device__ __forceinline__ void issue_chunk(Pipe &p, int slot_off, int piece = -1) {
  if (p.dma_left > 0) {
    if (p.wave < 6) {
#ifndef REFNERF_EXP_NODMA
      lptr_t dst = (lptr_t)(p.wbuf + slot_off + p.wave * 3072);
      if (piece < 0 || piece == 0) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 0, REFNERF_DMA_AUX);
      if (piece < 0 || piece == 1) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 1024, REFNERF_DMA_AUX);
      if ((piece < 0 || piece == 2) && p.wave < 5) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 2048, REFNERF_DMA_AUX);
#endif
    }
    if (piece >= 0 && piece < 2) return;             /* the stream position moves on with the last piece */
    p.src += BF_CHUNK_BYTES;
    if constexpr (SPLIT) {
      /* a pass streams [spatial section][spatial section][directional section] */
      p.seq += 1;
      if (p.seq == SPPACKED.sp_chunks) p.src -= (size_t)SPPACKED.sp_chunks * BF_CHUNK_BYTES;
      else if (p.seq == SPPACKED.chunks_per_pass) { p.src -= (size_t)SPPACKED.total_chunks * BF_CHUNK_BYTES; p.seq = 0; }
    } else {
      if (p.src == p.src_end) p.src -= (size_t)BFPACKED.chunks_per_pass * BF_CHUNK_BYTES;
    }
    p.dma_left -= 1;
  }
}

/* TIMING-ONLY builds (wrong results; scripts/build_main_variant.sh, docs/EXPERIMENTS.md section 10): -DREFNERF_EXP_NOFRAG takes the A
 * fragments as whatever their registers hold (no LDS fragment traffic, same MFMAs), -DREFNERF_EXP_NODMA
 * leaves the weight stream's LDS-DMA instructions out (the ring keeps what the first chunks brought). */
template <typename MM>
__device__ __forceinline__ typename MM::v8 lds_frag(const char *q) {
#ifdef REFNERF_EXP_NOFRAG
  typename MM::v8 u;                              /* whatever the registers hold: no instruction at all */
  asm volatile("" : "=v"(u) : "v"(q));
  return u;
#else
  return *reinterpret_cast<const typename MM::v8 *>(q);
#endif
}

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
 * REAL_L real), BF_BNLDS (8 over `bn` + 8 over LDS, REAL_L real).  SPLIT: the chunk belongs to the split-f16 image (its
 * stream position wraps differently: issue_chunk).
 * FIRST: the chunk opens a slice (accumulator starts from the bias piece).
 * `a` is the A-fragment ring; on entry it holds fragments 0..AF-1 of this chunk,
 * on exit those of the next one. */
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
/* `hook(k)`: VALU work of the caller placed behind the MFMA of step k (the split kernel runs the epilogue of the previous
 * slice there, in the issue gaps of this slice's matrix instructions) */
template <typename MM, int KIND, int REAL_L, bool FIRST, bool SPLIT = false, typename Hook = NoHook>
__device__ __forceinline__ void bf_chunk(Pipe &p, typename MM::v8 (&a)[AF], const v4uu (&in)[16], const v4uu (&bn)[8], v16f &acc, Hook &&hook = Hook()) {
  typedef typename MM::v8 v8mm;
  constexpr int KS = (KIND == BF_LDS8) ? 8 : 16;
  constexpr int L0 = (KIND == BF_BNLDS) ? 8 : 0;      /* first LDS step (for the LDS kinds) */
  static_assert(KS % AF == 0, "ring phase");
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
    hook(k);
    if (KIND == BF_LDS8 || KIND == BF_BNLDS) {
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
      RN_RENDEZVOUS();
#endif
      issue_chunk<SPLIT>(p, p.fil_off, REFNERF_BF_SPREAD ? 0 : -1);
    }
    if (REFNERF_BF_SPREAD) {
      if (k == REFNERF_BF_RDV + 2) issue_chunk<SPLIT>(p, p.fil_off, 1);
      if (k == REFNERF_BF_RDV + 4) issue_chunk<SPLIT>(p, p.fil_off, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ring_rotate(p);
}

/* A wave whose 32 samples of this pass are all past the end (the partly filled last pass of a workgroup: N = 192 puts 384
 * samples into 2 x 256) takes no part in the MLP, but keeps the weight stream going: it joins every rendezvous of the pass
 * and issues its DMA pieces.  The matrix pipe of its SIMD then belongs to the one active wave -- a half-filled pass costs
 * about two thirds of a full one instead of all of it. */
template <bool SPLIT = false>
__device__ __forceinline__ void idle_pass(Pipe &p) {
#pragma unroll 1
  for (int c = 0; c < (SPLIT ? SPPACKED.chunks_per_pass : BFPACKED.chunks_per_pass); ++c) {
    RN_RENDEZVOUS();
    issue_chunk<SPLIT>(p, p.fil_off);
    ring_rotate(p);
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

/* cycle stamp from the wave index / lane the kernel already holds (RN_STAMP re-derives both from threadIdx: two more live
 * registers, which the split kernel spills) */
#define RN_STAMPW(A, slot) do { asm volatile("; RNMARK " #slot); if ((A).prof && blockIdx.x == (gridDim.x >> 1) && lane == 0) (A).prof[wave * 32 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

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
  p.cur_off = 0; p.nxt_off = BF_CHUNK_BYTES; p.nx2_off = 2 * BF_CHUNK_BYTES; p.fil_off = (REFNERF_RING_SLOTS - 1) * BF_CHUNK_BYTES;
  p.dma_left = n_pass * BFPACKED.chunks_per_pass;
  p.lane = lane; p.wave = wave; p.h = h;
  p.t_vm = 0; p.t_bar = 0;
  RN_STAMPW(A, 0);
  issue_chunk(p, p.cur_off);                                 /* overlaps with the resampler */
  issue_chunk(p, p.nxt_off);
#if REFNERF_RING_SLOTS == 4
  issue_chunk(p, p.nx2_off);
#endif

  resample_phase<BF_NW, false>(A, reinterpret_cast<float *>(Xb), TD, NRM, ray0, wave, lane);   /* P0 */
  RN_STAMPW(A, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           /* chunks 0 and 1 have landed */
  RN_STAMPW(A, 2);

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
    /* which (ray, sample) this lane's column is: recomputed at every use (P1, P4, P6) from laundered inputs instead of being
     * carried -- spilled -- across the MLP phases */
    auto locate = [&](int &g, int &rl, int &si, bool &valid) {
      int col_l = col, pass_l = pass0;
      asm volatile("" : "+v"(col_l), "+s"(pass_l));
      g = pass_l + col_l;
      rl = g / N;
      si = g - rl * N;
      valid = (g < n_tot) && (ray0 + rl < A.R);
    };
    /* RINGPS: the rays whose last sample this pass brings are composited behind it (every wave meets the barrier) */
    auto pass_epilogue = [&]() {
      if constexpr (RINGPS) {
        __syncthreads();
        const int end = (pass0 + BT < n_tot) ? pass0 + BT : n_tot;
        /* opaque copies: see level_fwd_split */
        int td_o = (int)(TD - smem), xp_o = (int)(XP - smem), ps_o = (int)(PS - smem), nrm_o = (int)(NRM - smem), lane_e = lane, pass_e = pass0;
        int ntot_e = n_tot, ray0_e = ray0, wave_e = wave;
        asm volatile("" : "+s"(td_o), "+s"(xp_o), "+s"(ps_o), "+s"(nrm_o), "+v"(lane_e), "+s"(pass_e), "+s"(ntot_e), "+s"(ray0_e), "+s"(wave_e));
        composite_phase<BF_NW, true, NPS_EVAL, PSM>(A, smem + td_o, smem + xp_o, smem + ps_o, ntot_e, ray0_e, wave_e, lane_e, nullptr, smem + nrm_o,
                                                    pass_e / N, end / N);
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
      /* ONE address register for the twelve HD rows and one for the ray record (see level_fwd_split) */
      int g, rl, si; bool valid;
      locate(g, rl, si, valid);
      int ci = col, ro = (valid ? rl : 0) * 12;
      asm volatile("" : "+v"(ci), "+v"(ro));
      float v[3], gp[3], raw_dif[3], raw_tint[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v[i] = RY[ro + 6 + i];
        gp[i] = HD[(1 + i) * BT + ci];
        raw_dif[i] = HD[(5 + i) * BT + ci];
        raw_tint[i] = HD[(8 + i) * BT + ci];
      }
      sample_heads<ENC_FAST>(cfg, HD[0 * BT + ci], gp, HD[4 * BT + ci], raw_dif, raw_tint, v, sh);
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
        int g, rl, si; bool valid;
        locate(g, rl, si, valid);
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
        RN_STAMPW(A, 17);
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
        RN_STAMPW(A, 18);
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
      RN_STAMPW(A, 3 + phase * 8);

      /* layer 0 of the trunk: inputs from LDS (+ bottleneck registers for the dir MLP) -> R0 */
      if (phase == 0) bf_layer<MM, BF_LDS8, BF_IPE_REAL_KS>(p, ar, 0, R0, bn, R0);
      else bf_layer<MM, BF_BNLDS, BF_DIR_REAL_KS>(p, ar, 0, R0, bn, R0);
      RN_STAMPW(A, 4 + phase * 8);
      /* layers 1..7: A (R0->R1), B (R1->R0); the third A carries the skip input */
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const int second = (it == 2) ? (phase ? 2 : 1) : 0;
        bf_layer<MM, BF_REG, 0>(p, ar, second, R0, bn, R1);
        if (it < 3) bf_layer<MM, BF_REG, 0>(p, ar, 0, R1, bn, R0);
      }
      RN_STAMPW(A, 5 + phase * 8);
      if (phase == 0) {
        /* P3: heads: 4 bottleneck blocks stay in registers, the scalar block goes to LDS HD */
#pragma unroll
        for (int ob = 0; ob < 5; ++ob) {
          v16f acc;
          bf_slice<MM, BF_REG, 0>(p, ar, 0, R1, bn, acc);
          if (ob < 4) pack_acc<MM, false>(acc, bn[2 * ob], bn[2 * ob + 1]);
          else {
            int cb = col + 4 * h * BT;                   /* one laundered base: rows are immediate offsets from it */
            asm volatile("" : "+v"(cb));
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              int row = (rr & 3) + 8 * (rr >> 2) + 4 * h;
              if (row < HD_ROWS) HD[((rr & 3) + 8 * (rr >> 2)) * BT + cb] = acc[rr];
            }
          }
        }
        wave_sync();
        RN_STAMPW(A, 6);
      } else {
        /* rgb: one slice */
        v16f acc;
        bf_slice<MM, BF_REG, 0>(p, ar, 0, R1, bn, acc);
        float raw_rgb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(acc[i], n, 64);
        /* opaque copies taken HERE: the per-lane output addresses of P6 must not be computed at
         * the top of the pass and carried (spilled) across both MLP phases */
        int g_w, rl_w, si_w; bool valid;
        locate(g_w, rl_w, si_w, valid);
        int lane_w = lane, pass_w = pass0;
        asm volatile("" : "+v"(lane_w), "+s"(pass_w));
        if (valid && h == 0) {                                                            /* P6 */
          SampleHeads sh;
          load_heads(sh);
          colour_store<ENC_FAST, NPS_EVAL, PSM>(A, sh, raw_rgb, PS, PX, n_tot, g_w, col);
        }
        wave_sync();
        history_flush<NPS_EVAL, PSM>(A, PS, PX, n_tot, pass_w + wave * 32, wave * 32, (size_t)ray0 * N + pass_w + wave * 32, lane_w);
        RN_STAMPW(A, 14);
      }
    }
    __builtin_amdgcn_wave_barrier();
    pass_epilogue();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RN_STAMPW(A, 15);
#ifdef REFNERF_PROF_WAITS
  if (A.prof && blockIdx.x == 0 && lane == 0) { A.prof[wave * 32 + 20] = p.t_vm; A.prof[wave * 32 + 21] = p.t_bar; }
#endif

  if constexpr (!RINGPS) composite_phase<BF_NW, true, NPS_EVAL>(A, TD, XP, PS, n_tot, ray0, wave, lane, reinterpret_cast<float *>(WB), NRM);   /* P7 */
  RN_STAMPW(A, 16);
}

#ifndef REFNERF_SECONDARY_TU   /* (the second translation unit takes the device functions of this header, not its kernels) */
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_bf16(const LevelArgs A) { level_fwd_mm<MmBf16>(A); }
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16(const LevelArgs A) { level_fwd_mm<MmF16>(A); }
/* the same with the per-sample records in a ring (rays_per_wg * N > 640) */
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_bf16_ring(const LevelArgs A) { level_fwd_mm<MmBf16, true>(A); }
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16_ring(const LevelArgs A) { level_fwd_mm<MmF16, true>(A); }
#endif

/* =====================================================================================================================
 * REFNERF_PREC_F16X2 -- the parity-grade 16-bit mode (split operands, refnerf_layout.h "split-f16 operand image").
 *
 * Same skeleton as level_fwd_mm (8 waves, LDS-DMA chunk ring, one rendezvous per chunk, activations in registers), but
 *   * the spatial trunk runs TWICE per pass on 16 samples per wave, on v_mfma_f32_16x16x32_f16: a layer's input is an H and
 *     an L fragment set (hi / lo halves of 8 k-steps x 32 features), and W_hi H + W_lo H + W_hi L accumulate into the two
 *     16-row tiles of a 32-row slice (fp32; lo x lo = 2^-22 of a product is dropped).  The 8 values a lane holds after a slice
 *     are -- after ReLU and the hi / lo split -- its 8 B elements of the next layer's k-step: no cross-lane traffic in the
 *     epilogue, the weight image carries the feature permutation (refnerf_layout.h) -- 22 significand bits end to end;
 *   * the scalar head block (density, grad_pred, roughness, diffuse, tint) likewise; the bottleneck takes W_hi only;
 *   * the directional trunk is the plain f16 trunk over all 32 samples (the two runs' bottlenecks are merged with one
 *     v_permlane16_swap per dword);
 *   * everything outside the contractions is the fp32 parity code: bit-exact resampler (sequential CDF), libm-accurate
 *     encodings / activations / compositing.
 * MFMAs per wave and pass: 2 x 3128 of 16.5 cycles + 1100 of 32 = 138 k matrix cycles (plain kernel: 2272 x 32 = 73 k).
 * ===================================================================================================================== */
typedef unsigned v2uu __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_f16(float lo, float hi) { return cvt_pk_mm<MmF16>(lo, hi); }
/* (x0, x1) -> packed hi halves + packed lo halves, x = hi + lo.  The residual is formed from the BITS that get stored:
 * hipcc otherwise converts the same value twice with different instructions -- v_cvt_pk_f16_f32 of the fp32 product for the
 * stored half, v_fma_mixlo_f16 (the product rounded ONCE, to f16) for the copy the residual is taken from -- and the two
 * disagree by an ulp of the hi half for one value in ~10^5 (double rounding): 0.3 % of the samples then carried one IPE
 * feature that was off by 2^-11 (found with a per-layer dump build in round 3). */
__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned &hi, unsigned &lo) {
  hi = pk_f16(x0, x1);
  asm("" : "+v"(hi));
  const MmF16::v2 hv = __builtin_bit_cast(MmF16::v2, hi);
  const _Float16 h0 = hv[0], h1 = hv[1];
  lo = pk_f16(x0 - (float)h0, x1 - (float)h1);
}
/* ---- the spatial section on v_mfma_f32_16x16x32_f16: three partial products into one accumulator (refnerf_layout.h) ---- */
typedef _Float16 sq_v8 __attribute__((ext_vector_type(8)));
constexpr int SQ_NF = 4;                         /* fragment ring: the four pieces of a k-step */
struct SqAcc { v4f t0, t1; };                    /* the two 16-row tiles of a 32-row slice: lane (b, n) holds rows 4 b .. 4 b + 3 for sample n */
template <int KIND> constexpr int sq_nm() { return KIND == SQ_X ? 18 : (KIND == SQ_BN ? 32 : 24); }      /* MFMAs of a chunk */
template <int KIND> constexpr int sq_np() { return KIND == SQ_X ? 12 : 16; }                              /* pieces it consumes */
/* static schedule of MFMA j of a chunk: k-step, piece, hi / lo input fragment, tile, and the piece whose last use it is.
 * A / B / X, per k-step: [WhT0 H | WhT1 H | WlT0 H | WlT1 H | WhT0 L | WhT1 L] (per accumulator: hi*hi, lo*hi, hi*lo);
 * BN: [WhT0 H | WhT1 H | WhT0 L | WhT1 L];  SC: [WhT0 H | WlT0 H | WhT0 L] */
template <int KIND> constexpr int sq_step(int j) { return KIND == SQ_BN ? j / 4 : (KIND == SQ_SC ? j / 3 : j / 6); }
template <int KIND> constexpr int sq_piece(int j) {
  if (KIND == SQ_BN) return 2 * (j / 4) + (j & 1);
  if (KIND == SQ_SC) return 2 * (j / 3) + ((j % 3) == 1 ? 1 : 0);
  return 4 * (j / 6) + ((j % 6) < 4 ? (j % 6) : (j % 6) - 4);
}
template <int KIND> constexpr bool sq_lo(int j) { return KIND == SQ_BN ? (j & 3) >= 2 : (KIND == SQ_SC ? (j % 3) == 2 : (j % 6) >= 4); }
template <int KIND> constexpr int sq_tile(int j) { return KIND == SQ_SC ? 0 : (KIND == SQ_BN ? (j & 1) : ((j % 6) & 1)); }
template <int KIND> constexpr int sq_release(int j) {
  if (KIND == SQ_BN) return (j & 3) >= 2 ? 2 * (j / 4) + ((j & 3) - 2) : -1;
  if (KIND == SQ_SC) return (j % 3) == 1 ? 2 * (j / 3) + 1 : ((j % 3) == 2 ? 2 * (j / 3) : -1);
  const int jj = j % 6, sl = j / 6;
  return jj == 2 ? 4 * sl + 2 : (jj == 3 ? 4 * sl + 3 : (jj == 4 ? 4 * sl : (jj == 5 ? 4 * sl + 1 : -1)));
}
/* One chunk.  `fr` = the piece ring: on entry pieces 0..3 of this chunk, on exit those of the next one.  `in`: the layer input
 * as [H(s) L(s)] x 8 k-steps (SQ_X: from the LDS planes instead).  hook(j): caller's VALU work behind MFMA j. */
/* REFNERF_SQ_PREBIAS: a chunk that opens a slice finds its bias already in `acc` -- fetched by the chunk before it (PRE: this
 * chunk fetches the bias piece of the NEXT chunk into `nacc` right behind its rendezvous) -- instead of loading it and waiting
 * a full LDS round trip in front of its first MFMA */
#ifndef REFNERF_SQ_PREBIAS
#define REFNERF_SQ_PREBIAS 1
#endif
template <int KIND, bool FIRST, bool PRE, typename Hook = NoHook>
__device__ __forceinline__ void sq_chunk(Pipe &p, sq_v8 (&fr)[SQ_NF], const v4uu (&in)[16], SqAcc &acc, SqAcc &nacc, Hook &&hook = Hook()) {
  constexpr int NM = sq_nm<KIND>(), NP = sq_np<KIND>();
  constexpr int RDV = NM / 2 - 1;
  const char *w = p.wbuf + p.cur_off;
  const char *cur = w + 1024 + p.lane * 16;
  const char *nxt = p.wbuf + p.nxt_off + 1024 + p.lane * 16;
  sq_v8 xb[2];
  if (KIND == SQ_X) { xb[0] = lds_frag<MmF16>(p.xps); xb[1] = lds_frag<MmF16>(p.xps + (BT / 2) * 16); }
  if (FIRST && !REFNERF_SQ_PREBIAS) {
    const v4f *bp = reinterpret_cast<const v4f *>(w + (p.lane >> 4) * 16);      /* bias piece [T][b][4] */
    acc.t0 = bp[0];
    acc.t1 = bp[4];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    const int sl = sq_step<KIND>(j);
    sq_v8 b;
    if (KIND == SQ_X) b = xb[sq_lo<KIND>(j) ? 1 : 0];
    else b = __builtin_bit_cast(sq_v8, in[2 * ((KIND == SQ_B ? 4 : 0) + sl) + (sq_lo<KIND>(j) ? 1 : 0)]);
    if (sq_tile<KIND>(j)) acc.t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[sq_piece<KIND>(j) % SQ_NF], b, acc.t1, 0, 0, 0);
    else acc.t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[sq_piece<KIND>(j) % SQ_NF], b, acc.t0, 0, 0, 0);
    const int rel = sq_release<KIND>(j);
    if (rel >= 0) {
      const int q = rel + SQ_NF;
      fr[rel % SQ_NF] = (q < NP) ? lds_frag<MmF16>(cur + q * 1024) : lds_frag<MmF16>(nxt + (q - NP) * 1024);
    }
    hook(j);
    if (KIND == SQ_X) {
      /* the next k-step's H once this one's four H products are issued, its L behind the two L products */
      if ((j % 6) == 3 && sl + 1 < 3) xb[0] = lds_frag<MmF16>(p.xps + (sl + 1) * (4 * BT * 16));
      if ((j % 6) == 5 && sl + 1 < 3) xb[1] = lds_frag<MmF16>(p.xps + (sl + 1) * (4 * BT * 16) + (BT / 2) * 16);
    }
    if (j == RDV) {
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
      RN_RENDEZVOUS();
#endif
      issue_chunk<true>(p, p.fil_off, REFNERF_BF_SPREAD ? 0 : -1);
      if (PRE && REFNERF_SQ_PREBIAS) {
        const v4f *bp = reinterpret_cast<const v4f *>(p.wbuf + p.nxt_off + (p.lane >> 4) * 16);
        nacc.t0 = bp[0];
        nacc.t1 = bp[4];
      }
    }
    if (REFNERF_BF_SPREAD) {
      if (j == RDV + 4) issue_chunk<true>(p, p.fil_off, 1);
      if (j == RDV + 8) issue_chunk<true>(p, p.fil_off, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ring_rotate(p);
}
/* piece q (0..3) of a slice's epilogue: accumulator values (2 q, 2 q + 1) of the eight a lane holds -> NaN-propagating ReLU,
 * hi / lo split -> dword q of the next layer's H and L fragments of k-step `slice` */
__device__ __forceinline__ void sq_epi_piece(const SqAcc &a, int q, v4uu &oh, v4uu &ol) {
  const float y0 = q == 0 ? a.t0[0] : (q == 1 ? a.t0[2] : (q == 2 ? a.t1[0] : a.t1[2]));
  const float y1 = q == 0 ? a.t0[1] : (q == 1 ? a.t0[3] : (q == 2 ? a.t1[1] : a.t1[3]));
#ifdef REFNERF_SPLIT_RELU_MAX
  const float x0 = fmaxf(y0, 0.0f), x1 = fmaxf(y1, 0.0f);
#else
  const float x0 = (y0 < 0.0f) ? 0.0f : y0, x1 = (y1 < 0.0f) ? 0.0f : y1;
#endif
  unsigned hi, lo;
  split_pair_f16(x0, x1, hi, lo);
  oh[q] = hi;
  ol[q] = lo;
}
/* One spatial layer: slice ob's epilogue rides behind the first MFMAs of slice ob + 1 (all eight waves run the chunks in
 * lockstep: VALU work between two slices idles the matrix pipe of every SIMD) */
template <bool LAYER0>
__device__ __forceinline__ void sq_layer(Pipe &p, sq_v8 (&fr)[SQ_NF], SqAcc (&accs)[2], bool skip, const v4uu (&in)[16], v4uu (&out)[16]) {
  /* slice ob accumulates in accs[(ob + 1) & 1]: on entry accs[1] holds the bias of slice 0, on exit that of the first slice
   * behind this layer (REFNERF_SQ_PREBIAS) */
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    SqAcc &acc = accs[(ob + 1) & 1];
    SqAcc &prev = accs[ob & 1];
    auto hook = [&](int j) {
      if (ob == 0 || j >= 8 || (j & 1)) return;
      sq_epi_piece(prev, j >> 1, out[2 * ob - 2], out[2 * ob - 1]);
    };
    if constexpr (LAYER0) sq_chunk<SQ_X, true, true>(p, fr, in, acc, prev, hook);
    else {
      sq_chunk<SQ_A, true, false>(p, fr, in, acc, prev, hook);
      sq_chunk<SQ_B, false, true>(p, fr, in, acc, prev);
      if (skip) sq_chunk<SQ_X, false, true>(p, fr, in, acc, prev);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sq_epi_piece(accs[0], q, out[14], out[15]);
  __builtin_amdgcn_sched_barrier(0);
}
/* the bias piece of the chunk in the `cur` slot (the first chunk of a run / of the heads: nothing ran ahead to fetch it) */
__device__ __forceinline__ void sq_bias_now(const Pipe &p, SqAcc &acc) {
  if (!REFNERF_SQ_PREBIAS) return;
  const v4f *bp = reinterpret_cast<const v4f *>(p.wbuf + p.cur_off + (p.lane >> 4) * 16);
  acc.t0 = bp[0];
  acc.t1 = bp[4];
}

/* directional layer of the split kernel: the plain layer on the split kernel's DMA schedule */
template <typename MM, int KIND0, int REAL0>
__device__ __forceinline__ void dir_layer(Pipe &p, typename MM::v8 (&a)[AF], int second, const v4uu (&in)[16], const v4uu (&bn)[8], v4uu (&out)[16]) {
  /* (pipelining this plain epilogue into the next slice as in sq_layer measured no gain: 105.6 k -> 104.6 k cycles) */
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    v16f acc;
    bf_chunk<MM, KIND0, REAL0, true, true>(p, a, in, bn, acc);
    if constexpr (KIND0 == BF_REG) {
      if (second == 2) bf_chunk<MM, BF_BNLDS, BF_DIR_REAL_KS, false, true>(p, a, in, bn, acc);
    }
    pack_acc<MM, true>(acc, out[2 * ob], out[2 * ob + 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool RINGPS = false>
__device__ __forceinline__ void level_fwd_split(const LevelArgs &A) {
  typedef MmF16 MM;
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
  const int col = wave * 32 + n;                             /* this lane's sample column (directional phase) */

  Pipe p;
  p.src = reinterpret_cast<const char *>(A.packed) + wave * 3072 + lane * 16;
  p.src_end = nullptr;
  p.wbuf = WB;
  p.xp = Xb + (h * BT + col) * 16;
  /* IPE planes of a run: [k-group][plane (hi | lo)][128 columns = wave * 16 + sample][16 B].  Lane (b = lane / 16, n = lane % 16)
   * reads k-group 4 s + b of k-step s: hi plane, the lo plane (BT / 2) * 16 bytes behind */
  p.xps = Xb + ((lane >> 4) * BT + wave * 16 + (lane & 15)) * 16;
  p.seq = 0;
  p.cur_off = 0; p.nxt_off = BF_CHUNK_BYTES; p.nx2_off = 2 * BF_CHUNK_BYTES; p.fil_off = (REFNERF_RING_SLOTS - 1) * BF_CHUNK_BYTES;
  p.dma_left = n_pass * SPPACKED.chunks_per_pass;
  p.lane = lane; p.wave = wave; p.h = h;
  p.t_vm = 0; p.t_bar = 0;
  RN_STAMPW(A, 0);
#ifdef REFNERF_PROF_WAITS
  if (A.prof && blockIdx.x == (gridDim.x >> 1) && lane == 0) A.prof[wave * 32 + 22] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  issue_chunk<true>(p, p.cur_off);                           /* overlaps with the resampler */
  issue_chunk<true>(p, p.nxt_off);
#if REFNERF_RING_SLOTS == 4
  issue_chunk<true>(p, p.nx2_off);
#endif

  resample_phase<BF_NW, true>(A, reinterpret_cast<float *>(Xb), TD, NRM, ray0, wave, lane);   /* P0: bit-exact CDF */
  /* (the EXACT resampler leaves the ray geometry to its caller: park it here as the plain kernel's does) */
#pragma clang loop unroll(disable)
  for (int rl = wave; rl < rpw; rl += BF_NW) {
    const int ray = ray0 + rl;
    if (ray >= A.R) break;
    float *RYw = NRM + 8 + rl * 12;
    if (lane < 10) {
      const float val = lane < 3 ? A.rays.d_origins[(size_t)ray * 3 + lane]
                      : lane < 6 ? A.rays.d_directions[(size_t)ray * 3 + lane - 3]
                      : lane < 9 ? A.rays.d_viewdirs[(size_t)ray * 3 + lane - 6] : A.rays.d_radii[ray];
      RYw[lane] = val;
      const float dx = __shfl(val, 3, 64), dy = __shfl(val, 4, 64), dz = __shfl(val, 5, 64);
      if (lane == 0) NRM[rl] = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
  }
  RN_STAMPW(A, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           /* chunks 0 and 1 have landed */
  RN_STAMPW(A, 2);

#ifndef REFNERF_BF_NOPRIO
  if (wave >= BF_NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  v4uu R0[16], R1[16];
  /* one fragment ring for both sections: four pieces for the 16x16x32 spatial chunks; the plain directional chunks use its
   * first AF entries (the ring always holds the leading pieces of the chunk about to run) */
  sq_v8 ar[SQ_NF];
#pragma unroll
  for (int d = 0; d < SQ_NF; ++d) ar[d] = lds_frag<MM>(WB + 1024 + lane * 16 + d * 1024);

  for (int pass0 = 0; pass0 < n_tot; pass0 += BT) {
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    /* which (ray, sample) this lane's column is: recomputed where it is needed (P4, P6) from laundered inputs instead of
     * being carried -- spilled -- across the two spatial runs */
    auto locate = [&](int &g, int &rl, bool &valid) {
      int col_l = col, pass_l = pass0;
      asm volatile("" : "+v"(col_l), "+s"(pass_l));
      g = pass_l + col_l;
      rl = g / N;
      valid = (g < n_tot) && (ray0 + rl < A.R);
    };
    auto pass_epilogue = [&]() {
      if constexpr (RINGPS) {
        __syncthreads();
        const int end = (pass0 + BT < n_tot) ? pass0 + BT : n_tot;
        /* opaque copies: the compositing's per-ray / per-lane addresses must be formed HERE, not hoisted in front of the
         * pass loop and carried (spilled) across the MLP phases (ring variant: 44 B/lane of scratch in round 3) */
        int td_o = (int)(TD - smem), xp_o = (int)(XP - smem), ps_o = (int)(PS - smem), nrm_o = (int)(NRM - smem), lane_e = lane, pass_e = pass0;
        int ntot_e = n_tot, ray0_e = ray0, wave_e = wave;
        asm volatile("" : "+s"(td_o), "+s"(xp_o), "+s"(ps_o), "+s"(nrm_o), "+v"(lane_e), "+s"(pass_e), "+s"(ntot_e), "+s"(ray0_e), "+s"(wave_e));
        composite_phase<BF_NW, false, NPS_EVAL, PSM>(A, smem + td_o, smem + xp_o, smem + ps_o, ntot_e, ray0_e, wave_e, lane_e, nullptr, smem + nrm_o,
                                                     pass_e / N, end / N);
      }
    };
    {
      const int g0 = pass0 + wave * 32;
      if (g0 >= n_tot || ray0 + g0 / N >= A.R) { idle_pass<true>(p); pass_epilogue(); continue; }
    }
    auto load_heads = [&](SampleHeads &sh) {
      /* ONE address register for the twelve rows (HD sits past the 64 KB immediate range of the ds instructions: left to
       * itself hipcc keeps a base per row, hoists them out of the pass loop and spills them) */
      int g, rl; bool valid;
      locate(g, rl, valid);
      int ci = col, ro = (valid ? rl : 0) * 12;
      asm volatile("" : "+v"(ci), "+v"(ro));
      float v[3], gp[3], raw_dif[3], raw_tint[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v[i] = RY[ro + 6 + i];
        gp[i] = HD[(1 + i) * BT + ci];
        raw_dif[i] = HD[(5 + i) * BT + ci];
        raw_tint[i] = HD[(8 + i) * BT + ci];
      }
      sample_heads<false>(cfg, HD[0 * BT + ci], gp, HD[4 * BT + ci], raw_dif, raw_tint, v, sh);
    };

    /* P3 of a run: heads.  Bottleneck slices: hi weights over the split input -> packed f16.  Run 0 keeps them (bn0: the only
     * registers that live across run 1); run 1 merges the two runs with one v_permlane16_swap per dword into the 32-sample B
     * fragments of the directional trunk (refnerf_layout.h: the bottleneck k-steps of dir.0 / dir.4 are packed in the order
     * this leaves).  Scalar block: all three products, to LDS HD.  (Run 1's heads sit behind the run loop, not in it: what a
     * loop iteration assigns and a later one reads is live through the whole loop body for the register allocator -- with the
     * merged fragments assigned inside, 16 registers were parked in scratch across every trunk.) */
    v4uu bn0[4];
    auto heads = [&](auto RUN, v4uu (&bn)[8]) {
      constexpr int run = decltype(RUN)::value;
      SqAcc ha[2];
      sq_bias_now(p, ha[1]);
#pragma unroll
      for (int ob = 0; ob < 5; ++ob) {
        SqAcc &acc = ha[(ob + 1) & 1];
        if (ob < 4) {
          sq_chunk<SQ_BN, true, true>(p, ar, R1, acc, ha[ob & 1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned pk = e == 0 ? pk_f16(acc.t0[0], acc.t0[1]) : (e == 1 ? pk_f16(acc.t0[2], acc.t0[3])
                              : (e == 2 ? pk_f16(acc.t1[0], acc.t1[1]) : pk_f16(acc.t1[2], acc.t1[3])));
            if (run == 0) bn0[ob][e] = pk;
            else {
              const v2uu r = __builtin_amdgcn_permlane16_swap(bn0[ob][e], pk, false, false);
              bn[2 * ob][e] = r[0];
              bn[2 * ob + 1][e] = r[1];
            }
          }
        } else {
          sq_chunk<SQ_SC, true, false>(p, ar, R1, acc, ha[ob & 1]);
          const int bq = lane_v >> 4;
          int csl = wave * 32 + 16 * run + (lane_v & 15) + 4 * bq * BT;   /* one laundered base: rows are immediate offsets from it */
          asm volatile("" : "+v"(csl));
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (4 * bq + i < HD_ROWS) HD[i * BT + csl] = acc.t0[i];
        }
      }
      wave_sync();
    };
    typedef std::integral_constant<int, 0> Run0;
    typedef std::integral_constant<int, 1> Run1;

#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { R0[e] = (v4uu){0, 0, 0, 0}; R1[e] = (v4uu){0, 0, 0, 0}; }
      {
      /* P1 of run `phase`: the wave's samples 16 * phase + (lane & 15); four lanes per sample, each 24 of the 96 IPE
       * features: block hb = sin / "cos", degrees 8 qq .. 8 qq + 7 -- k-groups 6 hb + 3 qq + q, q = 0..2 */
      const int i16 = lane_v & 15, part = lane_v >> 4;
      const int hb = part >> 1, qq = part & 1;
      const int cs = wave * 32 + 16 * phase + i16;          /* pass column of this lane's sample */
      const int gs = pass0 + cs;
      const int rls = gs / N, sis = gs - rls * N;
      const bool vs = (gs < n_tot) && (ray0 + rls < A.R);
      float o[3], d[3];
      const float *ry = RY + (vs ? rls : 0) * 12;
#pragma unroll
      for (int i = 0; i < 3; ++i) { o[i] = ry[i]; d[i] = ry[3 + i]; }
      const float radius = ry[9];
      const float *td = TD + (vs ? rls : 0) * (N + 1);
      const float t0 = td[vs ? sis : 0], t1 = td[vs ? sis + 1 : 1];
      float lm[3], lv[3];
      cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
      if (cfg.disable_integration) { lv[0] = 0.0f; lv[1] = 0.0f; lv[2] = 0.0f; }        /* models.py:228-231 */
      char *xw = Xb + (wave * 16 + i16) * 16;
      RN_STAMPW(A, 17);
      /* two features per trip (one dword of the hi plane, one of the lo plane): the libm sine is long, keep ONE copy pair */
#pragma clang loop unroll(disable)
      for (int t = 0; t < 12; ++t) {
        unsigned whi, wlo;
        {
          float f[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int kk = 2 * t + u;                  /* 3 * (j - 8 qq) + b */
            const int jj = kk / 3, b3 = kk - 3 * jj;
            const float m = b3 == 0 ? lm[0] : (b3 == 1 ? lm[1] : lm[2]);
            const float v = b3 == 0 ? lv[0] : (b3 == 1 ? lv[1] : lv[2]);
            f[u] = ipe_feature_split(m, v, 8 * qq + jj, hb);
          }
          split_pair_f16(f[0], f[1], whi, wlo);
        }
        char *dst = xw + (6 * hb + 3 * qq + (t >> 2)) * BT * 16 + (t & 3) * 4;
        *reinterpret_cast<unsigned *>(dst) = whi;
        *reinterpret_cast<unsigned *>(dst + (BT / 2) * 16) = wlo;
      }
      }
      RN_STAMPW(A, 18);
      wave_sync();
      RN_STAMPW(A, 3 + phase * 4);
      /* the directional chunks keep AF fragments ahead: fetch the rest of the first spatial chunk's k-step (complete
       * since the rendezvous in the middle of the chunk before it) */
      if (phase == 0) {
#pragma unroll
        for (int d = AF; d < SQ_NF; ++d) ar[d] = lds_frag<MM>(p.wbuf + p.cur_off + 1024 + lane_v * 16 + d * 1024);
      }
      SqAcc accs[2];
      sq_bias_now(p, accs[1]);
      sq_layer<true>(p, ar, accs, false, R0, R0);
      RN_STAMPW(A, 4 + phase * 4);
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        sq_layer<false>(p, ar, accs, it == 2, R0, R1);
        if (it < 3) sq_layer<false>(p, ar, accs, false, R1, R0);
      }
      RN_STAMPW(A, 5 + phase * 4);
      if (phase == 0) {
        v4uu none[8];
        heads(Run0(), none);
        RN_STAMPW(A, 6);
      }
    }
    v4uu bn[8];
    heads(Run1(), bn);
    RN_STAMPW(A, 10);
#pragma unroll
    for (int e = 0; e < 16; ++e) { R0[e] = (v4uu){0, 0, 0, 0}; R1[e] = (v4uu){0, 0, 0, 0}; }
    {
    /* P4: head activations, reflection, IDE (k' = IDE index; half 0 real, half 1 imaginary) */
    /* (ring variant, round 6: the phase's half-wave index from a lane index formed here -- what depends on the kernel's entry
     *  value is hoisted in front of the pass loop and parked in scratch; the plain kernel keeps its proven allocation) */
    const int h4 = RINGPS ? (fresh_lane() >> 5) : h;
    char *xs = Xb + col * 16;
    {
      SampleHeads sh;
      load_heads(sh);
      float ide[40];
#pragma unroll
      for (int q = 36; q < 40; ++q) ide[q] = 0.0f;
      if (cfg.dir_enc == REFNERF_DIRENC_POSENC) posenc_eval<false, true>(sh.refd[0], sh.refd[1], sh.refd[2], h4, [&](int q, float val) { ide[q] = val; });
      else ide_eval<false>(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, h4, [&](int q, float val) { ide[q] = val; });
      if (h4 == 0) ide[36] = sh.dot;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        v8mm pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = (mm_t)ide[q * 8 + e];
        *reinterpret_cast<v8mm *>(xs + (5 * h4 + q) * BT * 16) = pk;
      }
    }
    wave_sync();
    RN_STAMPW(A, 11);
    v8mm (&ad)[AF] = reinterpret_cast<v8mm (&)[AF]>(ar);     /* the ring's first AF entries */
    dir_layer<MM, BF_BNLDS, BF_DIR_REAL_KS>(p, ad, 0, R0, bn, R0);
    RN_STAMPW(A, 12);
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      dir_layer<MM, BF_REG, 0>(p, ad, (it == 2) ? 2 : 0, R0, bn, R1);
      if (it < 3) dir_layer<MM, BF_REG, 0>(p, ad, 0, R1, bn, R0);
    }
    RN_STAMPW(A, 13);
    /* rgb: one slice */
    v16f acc;
    bf_chunk<MM, BF_REG, 0, true, true>(p, ad, R1, bn, acc);
    float raw_rgb[3];
    const int lane6 = RINGPS ? fresh_lane() : lane;
    const int n6 = RINGPS ? (lane6 & 31) : n, h6 = RINGPS ? (lane6 >> 5) : h;
#pragma unroll
    for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(acc[i], n6, 64);
    int g_w, rl_w; bool valid;
    locate(g_w, rl_w, valid);
    int lane_w = lane6, pass_w = pass0;
    asm volatile("" : "+v"(lane_w), "+s"(pass_w));
    if (valid && h6 == 0) {                                                           /* P6 */
      SampleHeads sh;
      load_heads(sh);
      colour_store<false, NPS_EVAL, PSM, true, RINGPS>(A, sh, raw_rgb, PS, PX, n_tot, g_w, RINGPS ? wave * 32 + n6 : col);
    }
    wave_sync();
    history_flush<NPS_EVAL, PSM>(A, PS, PX, n_tot, pass_w + wave * 32, wave * 32, (size_t)ray0 * N + pass_w + wave * 32, lane_w);
    RN_STAMPW(A, 14);
    }
    __builtin_amdgcn_wave_barrier();
    pass_epilogue();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RN_STAMPW(A, 15);
#ifdef REFNERF_PROF_WAITS
  if (A.prof && blockIdx.x == (gridDim.x >> 1) && lane == 0) {
    A.prof[wave * 32 + 20] = p.t_vm; A.prof[wave * 32 + 21] = p.t_bar;
    A.prof[wave * 32 + 23] = (long long)__builtin_amdgcn_s_memrealtime();     /* 100 MHz: with slots 0 and 15, the shader clock */
  }
#endif
  if constexpr (!RINGPS) composite_phase<BF_NW, false, NPS_EVAL>(A, TD, XP, PS, n_tot, ray0, wave, lane, reinterpret_cast<float *>(WB), NRM);   /* P7 */
  RN_STAMPW(A, 16);
}

#ifndef REFNERF_SECONDARY_TU
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16x2(const LevelArgs A) { level_fwd_split<false>(A); }
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_f16x2_ring(const LevelArgs A) { level_fwd_split<true>(A); }
#endif

/* ---------------- bf16 weight image ---------------- */
__device__ __forceinline__ int ipe_col_of_kprime(int kp) { return kp; }   /* LDS order = canonical IPE order */

}  // namespace rn
