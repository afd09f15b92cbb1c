/*
 * refnerf_level_f32.h -- fp32-MFMA level kernel (the parity mode).
 *
 * workgroup = 4 waves = RPW whole rays; each wave owns 32-sample blocks.  The
 * MLP runs transposed, D[out][sample] = W x X on v_mfma_f32_32x32x2_f32, so a
 * layer's accumulator registers ARE the next layer's B operands: activations
 * never leave the register file; only the encodings (IPE 96, dir-MLP input
 * 204) sit in LDS.
 */
#pragma once
#include "refnerf_level_common.h"

namespace rn {

constexpr int T_TILE = 128;  /* samples per pass: 4 waves x 32 */

/* ------------------------------------------------------------------ */
/* fp32 MFMA GEMM op on one 32-sample block                           */
/* ------------------------------------------------------------------ */

/* A fragments come through a buffer descriptor over the packed image: the
 * per-lane part of the address is one constant VGPR (lane*STRIDE*4), the
 * per-step part is an SGPR/immediate, so the 1000+ loads of an op cost no VALU
 * address arithmetic (flat global loads made hipcc precompute and spill
 * hundreds of 64-bit pointers). */
template <int NOB, int STRIDE>
__device__ __forceinline__ void load_a(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float (&a)[NOB]) {
  if constexpr (STRIDE == 8) {
    /* a k-step of an 8-block op is two 1 KB planes [q][lane][4 blocks]: each dwordx4 load is contiguous over the wave
     * (the lane-major [lane][8 blocks] form made every load touch twice the cache lines it used) */
    v4f x = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    a[0] = x[0];
    if constexpr (NOB > 1) { a[1] = x[1]; a[2] = x[2]; a[3] = x[3]; }
    if constexpr (NOB == 5) a[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + 1024, 0));
    if constexpr (NOB > 5) {
      v4f y = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + 1024, 0));
#pragma unroll
      for (int i = 4; i < NOB; ++i) a[i] = y[i - 4];
    }
  } else if constexpr (STRIDE == 4) {
    v4f x = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
#pragma unroll
    for (int i = 0; i < NOB; ++i) a[i] = x[i];
  } else {
    a[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
  }
}

/* A vector stored in accumulator layout [ob][h][16] (bias images, WD). */
template <int NOB>
__device__ __forceinline__ void load_acc(__amdgpu_buffer_rsrc_t rs, int off, int h, v16f (&out)[NOB]) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    v4f b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      b[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, h * 64 + q * 16, off * 4 + ob * 128, 0));
    out[ob] = (v16f){b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], b[1][2], b[1][3],
                     b[2][0], b[2][1], b[2][2], b[2][3], b[3][0], b[3][1], b[3][2], b[3][3]};
  }
}

/* out[ob] = bias + W * in, W from the packed image `rs`.  a_off/b_off: float
 * offsets of the op inside the image (wave-uniform); `xl` = LDS X + h*T_TILE +
 * column (for LDS steps).  The A stream is software-pipelined PF steps ahead
 * through a register ring; sched_barrier pins "MFMAs of step s, then the loads
 * of step s+PF" so that hipcc cannot sink the loads back to their uses (it
 * otherwise emits load; s_waitcnt vmcnt(0); mfma).  lds_steps % PF == 0. */
#ifndef REFNERF_PF
#define REFNERF_PF 3
#endif
constexpr int PF = REFNERF_PF;
/* cache policy of the activation / delta streams (written once, read by a later kernel): 2 = nt (streaming,
 * evict-first), so that 9 GB of them per launch do not push the 5 MB weight image out of the 4 MB L2s */
#ifndef REFNERF_STREAM_AUX
#define REFNERF_STREAM_AUX 2
#endif
struct NoStepHook {
  __device__ __forceinline__ void operator()(int, float) {}
  __device__ __forceinline__ void operator()(int) {}
  __device__ __forceinline__ void operator()(int, int) {}
};

/* Hook of gemm_op that streams the op's B operand (the 256 register values of this lane, i.e. a
 * layer input in the forward / a layer delta in the backward) to its rows of a [rows][pitch] fp32 matrix,
 * ONE store per k-step.  Issued as a burst of 128 stores at the layer boundary, the same bytes arrive
 * from all 1024 waves of the chip at the same moment (67 MB per layer at C2) and the next GEMM's first
 * A-fragment wait sits behind them (vmcnt counts loads and stores in order on gfx9): measured ~20 k
 * cycles of wait per layer in the backward.  Spread over the GEMM the stores ride under the MFMAs.
 * Rows follow the accumulator layout: row(r) = (r&3) + 8*(r>>2) (+4h in `voff`), i.e. +1,+1,+1,+5 rows
 * per step; the buffer descriptor is re-based every 32 rows so that 32-bit offsets suffice for any pitch. */
/* the value of lane 0 (all lanes of the chain kernels are active) */
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
/* H16: the matrix holds bf16 rows (the bf16-chain kernels: the values are bf16-exact, the stream is half as large) */
template <bool H16 = false>
struct RowStoreHookT {
  char *base;                 /* matrix + row0 * pitch (wave-uniform) */
  unsigned long long blk_bytes;   /* 32 rows */
  unsigned voff;              /* ((4h) * pitch + column) * 4, or 0xfffffff0 for lanes that must not store */
  unsigned p1, p5;            /* 1 and 5 rows in bytes */
  unsigned soff;
  __amdgpu_buffer_rsrc_t rs;
  static constexpr int ESZ = H16 ? 2 : 4;
  __device__ __forceinline__ RowStoreHookT(float *matrix, long long pitch, int row0, size_t col, int h, bool store) {
    /* the column of the wave's first lane goes into the 64-bit base: a blocked column (rb_col) does not fit 32 bits,
     * the distance to it inside a wave (at most two 64-sample blocks) does */
    const long long c0 = uniform64((long long)col);
    base = reinterpret_cast<char *>(matrix) + ((long long)row0 * pitch + c0) * ESZ;
    blk_bytes = (unsigned long long)pitch * 32ull * ESZ;
    voff = store ? (unsigned)(((long long)(4 * h) * pitch + ((long long)col - c0)) * ESZ) : 0xfffffff0u;
    p1 = (unsigned)(pitch * ESZ);
    p5 = 5u * p1;
    soff = 0;
    rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x80000000, 0x00020000);
  }
  __device__ __forceinline__ void operator()(int step, float b) {
    if ((step & 15) == 0 && step > 0) {
      rs = __builtin_amdgcn_make_buffer_rsrc(base + (unsigned long long)(step >> 4) * blk_bytes, 0, 0x80000000, 0x00020000);
      soff = 0;
    }
#ifndef REFNERF_EXPERIMENT_NO_STREAM   /* timing experiment only: drops the stream (wrong gradients) */
    if constexpr (H16) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(__builtin_bit_cast(unsigned, b) >> 16), rs, voff, soff, REFNERF_STREAM_AUX);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, b), rs, voff, soff, REFNERF_STREAM_AUX);
#endif
    soff += ((step & 3) == 3) ? p5 : p1;
  }
  /* the same unit walk with a raw dword (REFNERF_ACT_F16X2: units 2j / 2j + 1 = the packed hi / lo halves of rows 2j, 2j + 1,
   * i.e. step 8 t + e stores dword e >> 1 of the k-step's hi (e even) or lo (e odd) fragment -- no arithmetic, and no float
   * register in between: a packed pair of halves is not a well-formed float) */
  __device__ __forceinline__ void raw(int step, unsigned w) {
    static_assert(!H16, "raw dwords go to 4-byte units");
    if ((step & 15) == 0 && step > 0) {
      rs = __builtin_amdgcn_make_buffer_rsrc(base + (unsigned long long)(step >> 4) * blk_bytes, 0, 0x80000000, 0x00020000);
      soff = 0;
    }
#ifndef REFNERF_EXPERIMENT_NO_STREAM
    __builtin_amdgcn_raw_buffer_store_b32(w, rs, voff, soff, REFNERF_STREAM_AUX);
#endif
    soff += ((step & 3) == 3) ? p5 : p1;
  }
};
typedef RowStoreHookT<false> RowStoreHook;

/* The bf16-row variant: operator()(j, dword) stores the packed pair of features (accumulator registers 2j, 2j+1 of the
 * 128 a lane holds = rows row(2j), row(2j)+1) as one dword of pair-row row(2j)/2.  Pair-rows of a 32-row block: {0,1,4,5,
 * 8,9,12,13} + 2h, i.e. +1,+3 alternating; 16 pair-rows per block; j = 0..63. */
struct PairStoreHook {
  char *base;
  unsigned long long blk_bytes;
  unsigned voff, p1, p3, soff;
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ PairStoreHook(float *matrix, long long pitch, int row0, size_t col, int h, bool store) {
    const long long c0 = uniform64((long long)col);
    base = reinterpret_cast<char *>(matrix) + ((long long)(row0 >> 1) * pitch + c0) * 4;
    blk_bytes = (unsigned long long)pitch * 64ull;
    voff = store ? (unsigned)(((long long)(2 * h) * pitch + ((long long)col - c0)) * 4) : 0xfffffff0u;
    p1 = (unsigned)(pitch * 4);
    p3 = 3u * p1;
    soff = 0;
    rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x80000000, 0x00020000);
  }
  __device__ __forceinline__ void operator()(int j, unsigned dword) {
    if ((j & 7) == 0 && j > 0) {
      rs = __builtin_amdgcn_make_buffer_rsrc(base + (unsigned long long)(j >> 3) * blk_bytes, 0, 0x80000000, 0x00020000);
      soff = 0;
    }
#ifndef REFNERF_EXPERIMENT_NO_STREAM
    __builtin_amdgcn_raw_buffer_store_b32(dword, rs, voff, soff, REFNERF_STREAM_AUX);
#endif
    soff += (j & 1) ? p3 : p1;
  }
};

template <int NOB, int STRIDE, bool HAS_REG, bool BIAS = true, typename Hook = NoStepHook, int PF = rn::PF, bool ACC = false>
__device__ __forceinline__ void gemm_op(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h,
                                        const v16f (&in)[8], v16f (&out)[NOB], const float *xl,
                                        int lds_steps, Hook hook = Hook()) {
  constexpr int STEP_BYTES = 64 * STRIDE * 4;
  const int voff = lane * (STRIDE == 8 ? 16 : STRIDE * 4);
  int soff = a_off * 4;
  float a[PF][NOB];
#pragma unroll
  for (int d = 0; d < PF; ++d) load_a<NOB, STRIDE>(rs, voff, soff + d * STEP_BYTES, a[d]);
  soff += PF * STEP_BYTES;
  if constexpr (ACC) { /* `out` carries on (a further input group of the same layer) */ }
  else if constexpr (BIAS) load_acc<NOB>(rs, b_off, h, out);
  else {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = 0.0f;
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
      hook(step, b);
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

/* ------------------------------------------------------------------------------------------------
 * bf16 chains (cfg.precision = REFNERF_PREC_BF16 in refnerf_level_backward): the transposed GEMMs of the two
 * trunks, the two 201-row input blocks and the head block on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), deltas
 * rounded to bf16 once per layer.  Same kernel structure as the fp32 chains -- every wave streams its own A
 * fragments from L2 / L1 through a register ring, no LDS staging, no barriers: 1 KB per MFMA and wave is
 * 128 B/clk per CU at full MFMA rate, twice what the L1 delivers.  The image is laid out [k-step][block][lane][8 bf16]
 * so that every A-fragment load is one contiguous KB (a first [k-step][lane][block] layout touched 64 cache lines per
 * instruction and made the chains address-bound: 35 k cycles per layer).  Measured now: ~20 k cycles per chain layer
 * against 65 k for the fp32 chains (its 128 MFMAs need 4 k); a 2-step register ring is best (deeper rings spill), barriers
 * that keep the 4 waves in the same L1 window change nothing.
 * The head / rgb recompute and everything per sample stay fp32.
 * ------------------------------------------------------------------------------------------------ */
#ifndef REFNERF_PF16
#define REFNERF_PF16 4
#endif
constexpr int PF16 = REFNERF_PF16;
template <int NOB>
__device__ __forceinline__ void load_a16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, v8bf (&a)[NOB]) {
  /* image layout [k-step][ob][lane][8 bf16]: one instruction = 64 lanes x 16 B = 1 KB contiguous (8 cache lines); the
   * first layout, [k-step][lane][ob], made every instruction touch 64 lines (one per lane) and the chains TA-bound */
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
    a[ob] = __builtin_bit_cast(v8bf, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + ob * 1024, 0));
}

/* out[ob] = (bias +) W-block x B over REG_STEPS k-steps of packed register fragments `in` (a layer's input / delta in
 * accumulator order) followed by LDS_STEPS k-steps built from the fp32 LDS tile `xc` (= X + column), rows
 * 16 s + 8 h + e (encodings / the head block).  b_off: bias image of the fp32 ops (accumulator layout) when BIAS.
 * hook(step): runs once per register k-step (activation / delta stores). */
/* PFD: depth of the A register ring; a one-block op (NOB = 1) takes its whole stream up front (PFD = 16): at depth 2 its
 * 16 short k-steps paid an L2 latency every second step */
template <int NOB, int REG_STEPS16, int LDS_STEPS, bool BIAS, typename Hook = NoStepHook, int LDS_MAXROW = (1 << 30), int PFD = (NOB == 1 ? 16 : PF16)>
__device__ __forceinline__ void gemm_op_bf16(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h,
                                             const v4uu (&in)[16], v16f (&out)[NOB], const float *xc, Hook hook = Hook()) {
  constexpr int STEPS = REG_STEPS16 + LDS_STEPS;
  constexpr int STEP_BYTES = BT_STEP_FLOATS * 4;
  const int voff = lane * 16;
  int soff = a_off * 4;
  int hi = 128 * T_TILE;                         /* LDS rows >= 128: one laundered base (see tile_hi) */
  if constexpr (LDS_STEPS * 16 > 128) asm volatile("" : "+v"(hi));
  static_assert(PFD <= REG_STEPS16 + LDS_STEPS, "ring deeper than the op");
  v8bf a[PFD][NOB];
#pragma unroll
  for (int d = 0; d < PFD; ++d) load_a16<NOB>(rs, voff, soff + d * STEP_BYTES, a[d]);
  soff += PFD * STEP_BYTES;
  if constexpr (BIAS) load_acc<NOB>(rs, b_off, h, out);
  else {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    v8bf b;
    if (step >= REG_STEPS16) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int row = 16 * (step - REG_STEPS16) + 8 * h + e;
        if (row > LDS_MAXROW) row = LDS_MAXROW;          /* pad rows (zero weights) must still read finite values */
        x[e] = (16 * (step - REG_STEPS16) + 15 < 128) ? xc[row * T_TILE] : xc[(row - 128) * T_TILE + hi];
      }
      v4uu pk = {cvt_pk_bf16(x[0], x[1]), cvt_pk_bf16(x[2], x[3]), cvt_pk_bf16(x[4], x[5]), cvt_pk_bf16(x[6], x[7])};
      b = __builtin_bit_cast(v8bf, pk);
    } else {
      b = __builtin_bit_cast(v8bf, in[step < 16 ? step : 0]);
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
      out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step % PFD][ob], b, out[ob], 0, 0, 0);
    if (PFD <= PF16 || step + PFD < STEPS) load_a16<NOB>(rs, voff, soff + step * STEP_BYTES, a[step % PFD]);
    if (step < REG_STEPS16) hook(step);
    __builtin_amdgcn_sched_barrier(0);
  }
}

/* ------------------------------------------------------------------------------------------------
 * split-f16 chains (training forward with cfg.precision = REFNERF_PREC_F16X2): the same GEMMs with BOTH operands as hi + lo
 * pairs of IEEE halves, x = hi + lo (22 significand bits), products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with
 * fp32 accumulation -- the arithmetic of the parity-grade inference mode (refnerf_level_bf16.h) in the 4-wave training
 * skeleton: 32 samples per wave, the two halves of a layer input in 2 x 64 registers (one wave per SIMD: 512 registers),
 * three MFMAs per weight-fragment pair.  Every wave streams its own fragments ([k-step][hi | lo][ob][lane][8 halves], 16 KB
 * per k-step) through a register ring; the saved layer inputs leave as FP32 rows (hi + lo is exact in fp32), so the fp32
 * backward and the weight-gradient GEMM take them unchanged (REFNERF_ACT_F32).
 * ------------------------------------------------------------------------------------------------ */
typedef _Float16 v8hf __attribute__((ext_vector_type(8)));
typedef _Float16 v2hf __attribute__((ext_vector_type(2)));
/* (x0, x1) -> packed hi halves, packed lo halves; the residual is taken from the bits that are stored (hipcc otherwise
 * converts the same value twice with different roundings: see split_pair_f16 in refnerf_level_bf16.h) */
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned &hi, unsigned &lo) {
  v2hf hv = __builtin_convertvector((v2f){x0, x1}, v2hf);
  hi = __builtin_bit_cast(unsigned, hv);
  asm("" : "+v"(hi));
  hv = __builtin_bit_cast(v2hf, hi);
  const _Float16 h0 = hv[0], h1 = hv[1];
  const v2hf lv = __builtin_convertvector((v2f){x0 - (float)h0, x1 - (float)h1}, v2hf);
  lo = __builtin_bit_cast(unsigned, lv);
}
/* element e (0..7) of a packed hi / lo fragment pair back as fp32 (exact) */
__device__ __forceinline__ float split_elem(const v4uu &ph, const v4uu &pl, int e) {
  const unsigned wh = ph[e >> 1], wl = pl[e >> 1];
  const v2hf a = __builtin_bit_cast(v2hf, wh), b = __builtin_bit_cast(v2hf, wl);
  const _Float16 ah = (e & 1) ? a[1] : a[0], bl = (e & 1) ? b[1] : b[0];
  return (float)ah + (float)bl;
}
/* as gemm_op_bf16, on the split operands: `ih` / `il` = the packed hi / lo halves of the register k-steps; LDS k-steps are
 * split on the fly from the fp32 tile.  hook(step) once per register k-step.
 * ONE set of fragment registers (2 x NOB x 4): a block's hi / lo fragments of the next k-step are requested right behind
 * its three MFMAs of this one (24 MFMAs = 768 cycles per k-step cover the L2 round trip), the blocks go in pairs so that
 * no MFMA waits for the accumulator of its predecessor. */
template <int NOB, int REG_STEPS16, int LDS_STEPS, bool BIAS, typename Hook = NoStepHook, int LDS_MAXROW = (1 << 30), bool ACC = false>
__device__ __forceinline__ void gemm_op_split(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h,
                                              const v4uu (&ih)[16], const v4uu (&il)[16], v16f (&out)[NOB], const float *xc, Hook hook = Hook(),
                                              float lds_scale = 1.0f) {
  constexpr int STEPS = REG_STEPS16 + LDS_STEPS;
  constexpr int STEP_BYTES = 2 * BT_STEP_FLOATS * 4;
  const int voff = lane * 16;
  const int soff = a_off * 4;
  int hi = 128 * T_TILE;                         /* LDS rows >= 128: one laundered base (see tile_hi) */
  if constexpr (LDS_STEPS * 16 > 128) asm volatile("" : "+v"(hi));
  v8hf ah[NOB], al[NOB];
  auto fetch = [&](int ob, int step) {
    ah[ob] = __builtin_bit_cast(v8hf, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + step * STEP_BYTES + ob * 1024, 0));
    al[ob] = __builtin_bit_cast(v8hf, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + step * STEP_BYTES + BT_STEP_FLOATS * 4 + ob * 1024, 0));
  };
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) fetch(ob, 0);
  if constexpr (ACC) { /* `out` carries on (a further input group of the same layer) */ }
  else if constexpr (BIAS) load_acc<NOB>(rs, b_off, h, out);
  else {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    v8hf bh, bl;
    if (step >= REG_STEPS16) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int row = 16 * (step - REG_STEPS16) + 8 * h + e;
        if (row > LDS_MAXROW) row = LDS_MAXROW;          /* pad rows (zero weights) must still read finite values */
        x[e] = ((16 * (step - REG_STEPS16) + 15 < 128) ? xc[row * T_TILE] : xc[(row - 128) * T_TILE + hi]) * lds_scale;
      }
      v4uu ph, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned hh, ll;
        split_pair_h(x[2 * e], x[2 * e + 1], hh, ll);
        ph[e] = hh; pl[e] = ll;
      }
      bh = __builtin_bit_cast(v8hf, ph);
      bl = __builtin_bit_cast(v8hf, pl);
    } else {
      bh = __builtin_bit_cast(v8hf, ih[step < 16 ? step : 0]);
      bl = __builtin_bit_cast(v8hf, il[step < 16 ? step : 0]);
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ob += 2) {
      constexpr int dummy = 0; (void)dummy;
      const bool two = ob + 1 < NOB;
      out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ob], bh, out[ob], 0, 0, 0);
      if (two) out[ob + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ob + 1], bh, out[ob + 1], 0, 0, 0);
      out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ob], bh, out[ob], 0, 0, 0);
      if (two) out[ob + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ob + 1], bh, out[ob + 1], 0, 0, 0);
      out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ob], bl, out[ob], 0, 0, 0);
      if (two) out[ob + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ob + 1], bl, out[ob + 1], 0, 0, 0);
      /* (reads one step past the op at the end: the next op's data or the image's tail pad) */
#ifndef REFNERF_EXPERIMENT_NO_WSTREAM   /* timing experiment only: the operand stream stops after step 0 (wrong results) */
      fetch(ob, step + 1);
      if (two) fetch(ob + 1, step + 1);
#endif
      /* the hook's work (8 row stores + their hi + lo sums per k-step) in quarters behind the block pairs: at the end of the
       * step it ran after the last MFMA had issued, i.e. unhidden (one wave per SIMD) */
      if constexpr (NOB == 8) { if (step < REG_STEPS16) hook(step, ob >> 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NOB != 8) { if (step < REG_STEPS16) hook(step); }
    __builtin_amdgcn_sched_barrier(0);
  }
}

/* sample-major block of the bf16 ACT format (refnerf_layout.h SMB_*): slot `slot` of lane (gs, h) */
__device__ __forceinline__ v4u *smb_slot(const float *act, long long pitch, size_t gs, int h, int slot) {
  char *base = reinterpret_cast<char *>(const_cast<float *>(act)) + (size_t)SMB_ROW0 * (size_t)pitch * 4;
  return reinterpret_cast<v4u *>(base + ((((gs >> 5) * SMB_SLOTS + slot) * 64 + h * 32 + (gs & 31)) << 4));
}
__device__ __forceinline__ void smb_store(float *act, long long pitch, size_t gs, int h, int slot, v4u w) {
#ifndef REFNERF_EXPERIMENT_NO_STREAM
  __builtin_nontemporal_store(w, smb_slot(act, pitch, gs, h, slot));
#endif
}
__device__ __forceinline__ void smb_store_pk(float *act, long long pitch, size_t gs, int h, int slot0, const v4uu (&pk)[16]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) smb_store(act, pitch, gs, h, slot0 + t, (v4u){pk[t][0], pk[t][1], pk[t][2], pk[t][3]});
}
/* the block's x7 / v7 back as the fp32 accumulator image (bf16 values widened) */
__device__ __forceinline__ void smb_load_rows(const float *act, long long pitch, size_t gs, int h, int slot0, v16f (&x)[8]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const v4u w = *smb_slot(act, pitch, gs, h, slot0 + t);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[t >> 1][8 * (t & 1) + 2 * e] = __builtin_bit_cast(float, w[e] << 16);
      x[t >> 1][8 * (t & 1) + 2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
    }
  }
}

/* A 256 -> 256 chain layer with the weight stream SHARED by the four waves of the workgroup: per k-step every wave
 * fetches a quarter of the 8 KB image block (2 of the 8 row blocks: two 16-B loads per lane, two k-steps ahead), the
 * quarters meet in a two-slot LDS ring, and every wave reads its eight A fragments from there -- a quarter of the L2 -> L1
 * traffic of the per-wave stream (which ran at ~1250 cycles per k-step for 256 cycles of MFMA).  One s_barrier per
 * k-step, no memory fence: the history stores of the hook stay in flight across it (only this wave's ds_writes are
 * waited for).  Must be called by all four waves of the workgroup, the same number of times. */
#ifndef REFNERF_RING_FETCH
#define REFNERF_RING_FETCH 4   /* an L2 round trip under load outlasts two k-steps */
#endif
constexpr int RING_SLOTS = 3;
constexpr int RING_BYTES = RING_SLOTS * BT_STEP_FLOATS * 4;
/* BIAS: the accumulators start from the op's bias rows (forward); else from zero (backward: W^T delta) */
template <bool BIAS, typename Hook>
__device__ __forceinline__ void gemm_chain_bf16_shared(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h, int wave,
                                                       const v4uu (&in)[16], v16f (&out)[8], char *ring, Hook hook) {
  constexpr int STEPS = 16;
  constexpr int STEP_BYTES = BT_STEP_FLOATS * 4;                /* 8 KB: [ob][lane][8 bf16] */
  const int voff = wave * 2048 + lane * 16;
  const int soff = a_off * 4;
  char *wr = ring + wave * 2048 + lane * 16;
  const char *rd = ring + lane * 16;
  auto fetch = [&](int step, v4u (&g)[2]) {
    g[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + step * STEP_BYTES, 0);
    g[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 1024, soff + step * STEP_BYTES, 0);
  };
  auto stage = [&](int step, const v4u (&g)[2]) {
    *reinterpret_cast<v4u *>(wr + (step % RING_SLOTS) * STEP_BYTES) = g[0];
    *reinterpret_cast<v4u *>(wr + (step % RING_SLOTS) * STEP_BYTES + 1024) = g[1];
  };
  auto frags = [&](int step, v8bf (&a)[8]) {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) a[ob] = *reinterpret_cast<const v8bf *>(rd + (step % RING_SLOTS) * STEP_BYTES + ob * 1024);
  };
  auto rendezvous = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  /* step s: [fragments of s+1 <- slot (s+1)%3] [MFMAs of s] [stage s+2 -> slot (s+2)%3, fetch s+4] [rendezvous].
   * Slot (s+2)%3 was last read as step s-1, before the rendezvous of step s-1 that every wave has passed. */
  constexpr int GD = REFNERF_RING_FETCH;                        /* k-steps between a quarter's fetch and its LDS write */
  v4u g[GD][2];
  v8bf a[2][8];
#pragma unroll
  for (int d = 0; d < GD; ++d) fetch(d, g[d]);
  if constexpr (BIAS) load_acc<8>(rs, b_off, h, out);
  else {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = 0.0f;
  }
  stage(0, g[0]);
  fetch(GD, g[0]);
  stage(1, g[1]);
  fetch(GD + 1, g[1]);
  rendezvous();
  frags(0, a[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    if (step + 1 < STEPS) frags(step + 1, a[(step + 1) & 1]);
    const v8bf b = __builtin_bit_cast(v8bf, in[step]);
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
      out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step & 1][ob], b, out[ob], 0, 0, 0);
    if (step + 2 < STEPS) {
      stage(step + 2, g[(step + 2) % GD]);
      if (step + 2 + GD < STEPS) fetch(step + 2 + GD, g[(step + 2) % GD]);
    }
    hook(step);
    /* one wave per SIMD: everything else rides in the issue gaps of the 8 MFMAs -- the fragment reads of the next step
     * in the first four gaps (left behind the MFMAs they waited a full LDS round trip in front of the rendezvous),
     * then the stream's loads, its LDS writes and the history stores */
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
    if (step + 2 < STEPS) rendezvous();               /* the last two steps' slots are already complete */
    __builtin_amdgcn_sched_barrier(0);
  }
  /* the next call stages into slots 0 and 1 right away: their last readers (steps 15 and 13) must be done */
  rendezvous();
}

/* The same for the split-f16 chains.  A k-step of the split image is [8 hi fragments, 8 KB][8 lo fragments, 8 KB]; the ring
 * keeps its 8 KB slots and works in HALF steps: the even half step brings the hi fragments (16 MFMAs: hi * hi, hi * lo),
 * the odd one the lo fragments (8 MFMAs: lo * hi) -- gemm_op_split's products; an accumulator takes them as hi * hi, hi * lo,
 * lo * hi per k-step instead of hi * hi, lo * hi, hi * lo (fp32 accumulation: results agree to rounding, not bit for bit).  Per-wave streams pull 4 x 16 KB per k-step through the CU's 64 B/clk vector-memory path (1024 cycles
 * against 768 of matrix issue); shared, each wave fetches a quarter.  hook(step, quarter) as gemm_op_split's: quarters
 * 0, 1 ride in the even half step, 2, 3 in the odd one.  Must be called by all four waves, the same number of times. */
#ifndef REFNERF_SPLIT_SHARED
#define REFNERF_SPLIT_SHARED 0   /* measured (round 4, C2): forward 6.03 -> 6.20 ms, backward 3.65 -> 3.67: off (docs/EXPERIMENTS.md section 9) */
#endif
template <bool BIAS, typename Hook>
__device__ __forceinline__ void gemm_chain_split_shared(__amdgpu_buffer_rsrc_t rs, int a_off, int b_off, int lane, int h, int wave,
                                                        const v4uu (&ih)[16], const v4uu (&il)[16], v16f (&out)[8], char *ring, Hook hook) {
  constexpr int HSTEPS = 32;
  constexpr int HB = BT_STEP_FLOATS * 4;                        /* 8 KB: [ob][lane][8 f16] */
  const int voff = wave * 2048 + lane * 16;
  const int soff = a_off * 4;
  char *wr = ring + wave * 2048 + lane * 16;
  const char *rd = ring + lane * 16;
  auto fetch = [&](int u, v4u (&g)[2]) {
    g[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + u * HB, 0);
    g[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 1024, soff + u * HB, 0);
  };
  auto stage = [&](int u, const v4u (&g)[2]) {
    *reinterpret_cast<v4u *>(wr + (u % RING_SLOTS) * HB) = g[0];
    *reinterpret_cast<v4u *>(wr + (u % RING_SLOTS) * HB + 1024) = g[1];
  };
  auto frags = [&](int u, v8hf (&a)[8]) {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) a[ob] = *reinterpret_cast<const v8hf *>(rd + (u % RING_SLOTS) * HB + ob * 1024);
  };
  auto rendezvous = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  constexpr int GD = REFNERF_RING_FETCH;
  v4u g[GD][2];
  v8hf a[2][8];
#pragma unroll
  for (int d = 0; d < GD; ++d) fetch(d, g[d]);
  if constexpr (BIAS) load_acc<8>(rs, b_off, h, out);
  else {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[ob][r] = 0.0f;
  }
  stage(0, g[0]);
  fetch(GD, g[0]);
  stage(1, g[1]);
  fetch(GD + 1, g[1]);
  rendezvous();
  frags(0, a[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < HSTEPS; ++u) {
    const int st = u >> 1;
    if (u + 1 < HSTEPS) frags(u + 1, a[(u + 1) & 1]);
    const v8hf bh = __builtin_bit_cast(v8hf, ih[st]), bl = __builtin_bit_cast(v8hf, il[st]);
    if ((u & 1) == 0) {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ob], bh, out[ob], 0, 0, 0);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ob], bl, out[ob], 0, 0, 0);
    } else {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][ob], bh, out[ob], 0, 0, 0);
    }
    if (u + 2 < HSTEPS) {
      stage(u + 2, g[(u + 2) % GD]);
      if (u + 2 + GD < HSTEPS) fetch(u + 2 + GD, g[(u + 2) % GD]);
    }
    hook(st, 2 * (u & 1));
    hook(st, 2 * (u & 1) + 1);
    /* one wave per SIMD: the fragment reads of the next half step in the first gaps, then the stream's loads, its LDS
     * writes and the row stores */
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
    if (u + 2 < HSTEPS) rendezvous();                 /* the last two half steps' slots are already complete */
    __builtin_amdgcn_sched_barrier(0);
  }
  /* the next call stages into slots 0 and 1 right away: their last readers must be done */
  rendezvous();
}

/* ReLU and its sign bit without a compare: as integers, x > 0 <=> max_i32(x, 0) != 0 (negative floats and -0 are
 * negative integers), so relu(x) = max_i32(x, 0) and the mask bit = min_u32(relu(x), 1) -- VALU only (see keep_if_bit),
 * bit-identical to `x > 0 ? x : 0` for every non-NaN x. */
/* NANPROP (split-f16 chains): a NaN of either sign passes -- as signed integers the negative NaNs are the ones above -inf's
 * pattern -- so that a unit beyond the range of an IEEE half (hi = inf, lo = -inf -> NaN accumulators) reaches the outputs as
 * NaN instead of vanishing (include/refnerf_hip.h: REFNERF_PREC_F16X2 range) */
template <bool NANPROP = false>
__device__ __forceinline__ float relu_bit(float x, unsigned &mk, int bit) {
  const int xi = __builtin_bit_cast(int, x);
  const unsigned v = NANPROP ? (unsigned)(xi > (int)0xff800000 ? xi : 0) : (unsigned)(xi > 0 ? xi : 0);   /* (-inf = 0xff800000) */
  mk |= (v < 1u ? v : 1u) << bit;
  return __builtin_bit_cast(float, v);
}
/* x where bit `bit` of `mk` is set, else +0: v_bfe_i32 (0 / all ones) + v_and -- VALU only.  As `bit ? x : 0.0f` the
 * compiler produced 128 v_cmp results in SGPR pairs first, spilled them through v_writelane (+ s_nop hazards) and
 * selected afterwards: ~6 instructions per element instead of 2. */
__device__ __forceinline__ float keep_if_bit(float x, unsigned mk, int bit) {
  const int m = __builtin_amdgcn_sbfe((int)mk, bit, 1);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (unsigned)m);
}
/* delta through a ReLU (recorded mask) and straight into the next GEMM's packed B fragments */
__device__ __forceinline__ void mask_pack(const v16f (&out)[8], const unsigned (&mk)[4], v4uu (&pk)[16]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = keep_if_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pk[2 * ob][e] = cvt_pk_bf16(v[2 * e], v[2 * e + 1]);
      pk[2 * ob + 1][e] = cvt_pk_bf16(v[8 + 2 * e], v[8 + 2 * e + 1]);
    }
  }
}
/* element (blk, r) of a packed delta as fp32 (what the weight-gradient GEMM reads back from DELTA) */
__device__ __forceinline__ float pk_elem(const v4uu (&pk)[16], int blk, int r) {
  const unsigned w = pk[2 * blk + (r >> 3)][(r & 7) >> 1];
  return __builtin_bit_cast(float, (r & 1) ? (w & 0xffff0000u) : (w << 16));
}

/* forward: ReLU, its sign pattern (bit 16 (ob & 1) + r of mk[ob >> 1]) and the packed bf16 input of the next layer */
__device__ __forceinline__ void relu_mask_pack(const v16f (&out)[8], unsigned (&mk)[4], v4uu (&pk)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) mk[q] = 0u;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = relu_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pk[2 * ob][e] = cvt_pk_bf16(v[2 * e], v[2 * e + 1]);
      pk[2 * ob + 1][e] = cvt_pk_bf16(v[8 + 2 * e], v[8 + 2 * e + 1]);
    }
  }
}

__device__ __forceinline__ void relu_into(const v16f (&out)[8], v16f (&in)[8]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) in[ob][r] = fmaxf(out[ob][r], 0.0f);
}

/* forward: ReLU, its sign pattern and the packed hi / lo input of the next layer */
__device__ __forceinline__ void relu_mask_split(const v16f (&out)[8], unsigned (&mk)[4], v4uu (&ph)[16], v4uu (&pl)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) mk[q] = 0u;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = relu_bit<true>(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, l0, h1, l1;
      split_pair_h(v[2 * e], v[2 * e + 1], h0, l0);
      split_pair_h(v[8 + 2 * e], v[8 + 2 * e + 1], h1, l1);
      ph[2 * ob][e] = h0; pl[2 * ob][e] = l0;
      ph[2 * ob + 1][e] = h1; pl[2 * ob + 1][e] = l1;
    }
  }
}
/* Power-of-two factor that brings a sample's largest |value| m into [2^7, 2^8): IEEE halves span 2^-24 .. 2^16, and the
 * deltas of a backward chain are 1e-4 .. 1e-9 -- unscaled, their lo halves (and soon the hi halves) underflow and the
 * gradient dies down the chain (measured: 1.0 relative error at the first directional layers).  Columns of the B operand
 * are independent in W^T delta, so every SAMPLE carries its own factor through the chain. */
/* `live` (optional): whether the sample has any non-zero value at all -- a sample without gradient keeps factor 1, which
 * must not be mistaken for "the sample with the largest deltas" when the layer's smallest factor is taken (refnerf_wgrad_f16.h) */
__device__ __forceinline__ float pow2_scale_for(float m, bool *live = nullptr) {
  m = fmaxf(m, __shfl_xor(m, 32, 64));                    /* both half-waves hold values of the same sample */
  int e = 8 - __builtin_amdgcn_frexp_expf(m);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  if (live) *live = m > 0.0f;
  return (m > 0.0f) ? __builtin_ldexpf(1.0f, e) : 1.0f;
}
/* delta through a recorded ReLU mask into the next transposed GEMM's packed hi / lo fragments, rescaled per sample:
 * on entry `out` carries the factor `c`, on exit the fragments carry the updated `c` (DELTA rows are stored as value / c) */
__device__ __forceinline__ void mask_split(const v16f (&out)[8], const unsigned (&mk)[4], v4uu (&ph)[16], v4uu (&pl)[16], float &c, bool *live = nullptr) {
  float m = 0.0f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(keep_if_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r)));
  const float rs = pow2_scale_for(m, live);
  c *= rs;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = keep_if_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r) * rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, l0, h1, l1;
      split_pair_h(v[2 * e], v[2 * e + 1], h0, l0);
      split_pair_h(v[8 + 2 * e], v[8 + 2 * e + 1], h1, l1);
      ph[2 * ob][e] = h0; pl[2 * ob][e] = l0;
      ph[2 * ob + 1][e] = h1; pl[2 * ob + 1][e] = l1;
    }
  }
}


/* rows [row0 + 32*blk + row(r,h)] of a [rows][pitch] matrix, column gs: 128 B
 * contiguous per (row, half-wave).  Uniform 64-bit row base + 32-bit lane offset. */
__device__ __forceinline__ void stream_store(float *p, float v) {
#if REFNERF_STREAM_AUX
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
/* Element (row, col) of a [rows][pitch] matrix: fp32 rows, or (H16) bf16 rows stored in PAIRS -- rows 2j and 2j+1 share
 * the dwords of pair-row j (low / high half), so that a lane's two adjacent features (exactly a packed B-fragment dword)
 * leave in one 4-byte store and a half-wave writes a full 128-B line segment. */
template <bool H16>
__device__ __forceinline__ long long elem_index(int row, long long col, long long pitch) {
  if constexpr (H16) return ((long long)(row >> 1) * pitch + col) * 2 + (row & 1);
  else return (long long)row * pitch + col;
}
template <bool H16>
__device__ __forceinline__ void stream_store_e(float *base, long long idx, float v) {
  if constexpr (H16) {
    unsigned short *q = reinterpret_cast<unsigned short *>(base) + idx;
    const unsigned short w = __builtin_bit_cast(unsigned short, (__bf16)v);
#if REFNERF_STREAM_AUX
    __builtin_nontemporal_store(w, q);
#else
    *q = w;
#endif
  } else stream_store(base + idx, v);
}
template <bool H16>
__device__ __forceinline__ float load_e(const float *base, long long idx) {
  if constexpr (H16) return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(base)[idx] << 16);
  else return base[idx];
}
template <int NB, bool H16 = false>
__device__ __forceinline__ void store_rows(float *base, long long pitch, int row0, size_t gs, int h, bool valid, const v16f *x) {
  /* one 64-bit origin per lane, compile-time row offsets on top (row0 is even: the pair-row of row0 + 4h + c is that of
   * row0 + 4h plus c / 2, its half c & 1) */
  const long long e0 = (H16 ? (long long)((row0 + 4 * h) >> 1) : (long long)(row0 + 4 * h)) * pitch + (long long)gs;
  if (valid) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = blk * 32 + (r & 3) + 8 * (r >> 2);
        if constexpr (H16) stream_store_e<true>(base, (e0 + (long long)(c >> 1) * pitch) * 2 + (c & 1), x[blk][r]);
        else stream_store_e<false>(base, e0 + (long long)c * pitch, x[blk][r]);
      }
  }
}
template <bool H16 = false>
__device__ __forceinline__ void store_row1(float *base, long long pitch, int row, size_t gs, float v) {
  stream_store_e<H16>(base, elem_index<H16>(row, (long long)gs, pitch), v);
}

/* the same rows read back (the accumulator-layout image of a saved activation block) */
template <int NB, bool H16 = false>
__device__ __forceinline__ void load_rows(const float *base, long long pitch, int row0, size_t gs, int h, v16f *x) {
  const long long e0 = (H16 ? (long long)((row0 + 4 * h) >> 1) : (long long)(row0 + 4 * h)) * pitch + (long long)gs;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = blk * 32 + (r & 3) + 8 * (r >> 2);
      if constexpr (H16) {
        if ((r & 1) == 0) {                        /* registers r, r + 1 are the two halves of one pair-row dword */
          const unsigned w = reinterpret_cast<const unsigned *>(base)[e0 + (long long)(c >> 1) * pitch];
          x[blk][r] = __builtin_bit_cast(float, w << 16);
          x[blk][r + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
        }
      } else x[blk][r] = load_e<false>(base, e0 + (long long)c * pitch);
    }
}

/* ---- REFNERF_ACT_F16X2 (refnerf_layout.h): rows in pairs, unit 2j = packed hi halves of rows (2j, 2j + 1), unit 2j + 1 = packed
 * lo halves.  `row` even. ---- */
__device__ __forceinline__ void stream_store_u(float *base, long long idx, unsigned w) {
#if REFNERF_STREAM_AUX
  __builtin_nontemporal_store(w, reinterpret_cast<unsigned *>(base) + idx);
#else
  reinterpret_cast<unsigned *>(base)[idx] = w;
#endif
}
__device__ __forceinline__ void store_pair_split(float *base, long long pitch, int row, size_t gs, float x0, float x1) {
  unsigned hi, lo;
  split_pair_h(x0, x1, hi, lo);
  const long long e0 = (long long)row * pitch + (long long)gs;
  stream_store_u(base, e0, hi);
  stream_store_u(base, e0 + pitch, lo);
}
/* NB accumulator blocks (rows row0 + 32 blk + row(r, h)) as hi / lo pair units */
template <int NB>
__device__ __forceinline__ void store_rows_split(float *base, long long pitch, int row0, size_t gs, int h, bool valid, const v16f *x) {
  const long long e0 = (long long)(row0 + 4 * h) * pitch + (long long)gs;
  if (valid) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int c = blk * 32 + (r & 3) + 8 * (r >> 2);
        unsigned hi, lo;
        split_pair_h(x[blk][r], x[blk][r + 1], hi, lo);
        stream_store_u(base, e0 + (long long)c * pitch, hi);
        stream_store_u(base, e0 + (long long)(c + 1) * pitch, lo);
      }
  }
}
/* ... and read back as the fp32 accumulator image (hi + lo is exact in fp32) */
template <int NB>
__device__ __forceinline__ void load_rows_split(const float *base, long long pitch, int row0, size_t gs, int h, v16f *x) {
  const long long e0 = (long long)(row0 + 4 * h) * pitch + (long long)gs;
  const unsigned *u = reinterpret_cast<const unsigned *>(base);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int c = blk * 32 + (r & 3) + 8 * (r >> 2);
      const unsigned wh = u[e0 + (long long)c * pitch], wl = u[e0 + (long long)(c + 1) * pitch];
      const v2hf a = __builtin_bit_cast(v2hf, wh), b = __builtin_bit_cast(v2hf, wl);
      const _Float16 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
      x[blk][r] = (float)a0 + (float)b0;
      x[blk][r + 1] = (float)a1 + (float)b1;
    }
}

/* ... or as the packed hi / lo fragments of the 16 k-steps themselves (what the forward's GEMM consumed: dword e of k-step t = the
 * pair units of accumulator rows 2 e, 2 e + 1 of that k-step's block) */
__device__ __forceinline__ void load_frags_split(const float *base, long long pitch, int row0, size_t gs, int h, v4uu (&fh)[16], v4uu (&fl)[16]) {
  const long long e0 = (long long)(row0 + 4 * h) * pitch + (long long)gs;
  const unsigned *u = reinterpret_cast<const unsigned *>(base);
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 8 * (t & 1) + 2 * e, c = 32 * (t >> 1) + (r & 3) + 8 * (r >> 2);
      fh[t][e] = u[e0 + (long long)c * pitch];
      fl[t][e] = u[e0 + (long long)(c + 1) * pitch];
    }
}

/* ReLU that also records the sign pattern: bit (16*(ob&1) + r) of mk[ob>>1]. */
__device__ __forceinline__ void relu_mask_into(const v16f (&out)[8], v16f (&in)[8], unsigned (&mk)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) mk[q] = 0u;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      in[ob][r] = relu_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r);          /* relu'(0) = 0, as torch */
    }
}
/* in = out where the recorded ReLU was active, else 0 (delta through a ReLU). */
__device__ __forceinline__ void masked_into(const v16f (&out)[8], v16f (&in)[8], const unsigned (&mk)[4]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) in[ob][r] = keep_if_bit(out[ob][r], mk[ob >> 1], 16 * (ob & 1) + r);
}

/* Contribution of this lane's 48 IPE-gradient rows to d(raw_density)/d(lifted
 * mean) (coord.py:119-126 differentiated: d/dx [e*sin(r(x*2^j))] = e*cos(r)*2^j,
 * the variance path is detached with the rest of the sample geometry). */
__device__ __forceinline__ void ipe_vjp_accum(const v16f (&gi)[3], const float lm[3], const float lv[3], int h, float gl[3]) {
#pragma unroll
  for (int blk = 0; blk < 3; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int cb = k >= 48, kk = cb ? k - 48 : k;
      const int j = kk / 3, b = kk - 3 * j;
      const float m = (b == 0) ? lm[0] : (b == 1 ? lm[1] : lm[2]);
      const float vv = (b == 0) ? lv[0] : (b == 1 ? lv[1] : lv[2]);
      const float sc = __builtin_ldexpf(1.0f, j), sc2 = __builtin_ldexpf(1.0f, 2 * j);
      float x = m * sc;
      if (cb) x = x + HALF_PI_F;
      const float e = expf(-0.5f * (vv * sc2));
      const float t = ((gi[blk][r] * e) * cosf(safe_arg(x))) * sc;
      if (b == 0) gl[0] += t; else if (b == 1) gl[1] += t; else gl[2] += t;
    }
}

/* The same from the (e sin, e cos) bf16 pairs P1 left in tile rows 128 + kk (bf16 chains): d/dm [e sin(m 2^j)] =
 * (e cos) 2^j, d/dm [e cos(m 2^j)] = -(e sin) 2^j -- 48 LDS reads and ~5 VALU each instead of 48 expf + cosf with
 * range reduction (2 x 20 k cycles per pass).  Row k = c + 4h of a lane: sin / cos row by c alone (no c in 44..47),
 * (j, b) of kk = c' + 4h one of two compile-time pairs; a half-wave-1 lane's b is (c' + 1) % 3: rotated at the end. */
__device__ __forceinline__ void ipe_vjp_accum_lds(const v16f (&gi)[3], const float *X, int col, int h, float gl[3]) {
  int base = (BNECK * T_TILE + col + 4 * h * T_TILE) * 2;        /* ushort index of row 128 + 4h, this column */
  asm volatile("" : "+v"(base));
  const unsigned short *xs = reinterpret_cast<const unsigned short *>(X) + base;
  float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int blk = 0; blk < 3; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = blk * 32 + (r & 3) + 8 * (r >> 2);
      const int cb = c >= 48, cc = cb ? c - 48 : c;                /* cos row?  kk = cc + 4h */
      const float sc0 = (float)(1 << (cc / 3)), sc1 = (float)(1 << ((cc + 4) / 3));
      const float f = __builtin_bit_cast(float, (unsigned)xs[(cc * T_TILE) * 2 + (cb ? 0 : 1)] << 16);
      const float sc = h ? sc1 : sc0;
      acc[cc % 3] += (gi[blk][r] * f) * (cb ? -sc : sc);
    }
#pragma unroll
  for (int b = 0; b < 3; ++b) gl[b] += h ? acc[(b + 2) % 3] : acc[b];
}

/* Density-gradient normals (models.py:603-609): VJP of raw_density through the
 * spatial MLP (transposed packed ops, recorded ReLU masks) and the IPE, then
 * -normalize.  `in`/`out` are scratch; M[l] = mask of layer l (consumed). */
__device__ __forceinline__ void density_normals(__amdgpu_buffer_rsrc_t rs, int lane, int h, v16f (&in)[8], v16f (&out)[8],
                                                unsigned (&M)[8][4], const float lm[3], const float lv[3],
                                                const float *xl, float nrm_out[3]) {
  load_acc<8>(rs, PACKED.wd_off, h, out);
  masked_into(out, in, M[7]);
  float gl[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int i = 7; i >= 0; --i) {
    if (i == 5 || i == 0) {
      v16f gi[3];
      gemm_op<3, 4, true, false>(rs, PACKED.top[i == 5 ? TOP_SP5_IPE : TOP_SP0].a_off, 0, lane, h, in, gi, xl, 0);
      ipe_vjp_accum(gi, lm, lv, h, gl);
    }
    if (i > 0) {
      gemm_op<8, 8, true, false>(rs, PACKED.top[i - 1].a_off, 0, lane, h, in, out, xl, 0);
#pragma unroll
      for (int l = 7; l > 0; --l)
#pragma unroll
        for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
      masked_into(out, in, M[7]);
    }
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) gl[b] += __shfl_xor(gl[b], 32, 64);
  const float gx[3] = {-gl[2], -gl[1], -gl[0]};            /* basis^T (octahedron/1: lifted = (-z,-y,-x)) */
  const float ng = sqrtf(fmaxf((gx[0] * gx[0] + gx[1] * gx[1]) + gx[2] * gx[2], EPS32));
#pragma unroll
  for (int b = 0; b < 3; ++b) nrm_out[b] = -(gx[b] / ng);
}


/* The same for a general IPE basis (G groups of three directions): each group's 96 IPE-gradient rows come from its own
 * transposed block (group 0: the canonical TOP ops, groups 1..: the image tail), go through the IPE with the group's lifted
 * mean / variance (`lift(g, lm, lv)`), and the three per-direction derivatives are carried back to world space with the
 * basis rows (the transpose of coord.py:131's `mean @ basis`). */
template <typename Lift>
__device__ __forceinline__ void density_normals_gb(__amdgpu_buffer_rsrc_t rs, int lane, int h, v16f (&in)[8], v16f (&out)[8],
                                                   unsigned (&M)[8][4], const float *xl, int groups, const float *basis, Lift &&lift,
                                                   float nrm_out[3]) {
  load_acc<8>(rs, PACKED.wd_off, h, out);
  masked_into(out, in, M[7]);
  float gw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int i = 7; i >= 0; --i) {
    if (i == 5 || i == 0) {
#pragma unroll 1
      for (int gq = 0; gq < groups; ++gq) {
        v16f gi[3];
        const int a_off = gq == 0 ? PACKED.top[i == 5 ? TOP_SP5_IPE : TOP_SP0].a_off : pext_t_off(i == 5 ? 1 : 0, gq);
        gemm_op<3, 4, true, false>(rs, a_off, 0, lane, h, in, gi, xl, 0);
        float lm[3], lv[3], gl[3] = {0.0f, 0.0f, 0.0f};
        lift(gq, lm, lv);
        ipe_vjp_accum(gi, lm, lv, h, gl);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) gw[c] += gl[b] * basis[9 * gq + 3 * b + c];
      }
    }
    if (i > 0) {
      gemm_op<8, 8, true, false>(rs, PACKED.top[i - 1].a_off, 0, lane, h, in, out, xl, 0);
#pragma unroll
      for (int l = 7; l > 0; --l)
#pragma unroll
        for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
      masked_into(out, in, M[7]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) gw[c] += __shfl_xor(gw[c], 32, 64);
  const float ng = sqrtf(fmaxf((gw[0] * gw[0] + gw[1] * gw[1]) + gw[2] * gw[2], EPS32));
#pragma unroll
  for (int c = 0; c < 3; ++c) nrm_out[c] = -(gw[c] / ng);
}

/* density_normals on the bf16 chains: same VJP, deltas rounded to bf16 once per layer */
__device__ __forceinline__ void density_normals_bf16(__amdgpu_buffer_rsrc_t rs, int lane, int h, int wave, char *ring, v16f (&out)[8], v4uu (&pk)[16],
                                                     unsigned (&M)[8][4], const float *X, int col, float nrm_out[3]) {
  load_acc<8>(rs, PACKED.wd_off, h, out);
  mask_pack(out, M[7], pk);
  float gl[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int i = 7; i >= 0; --i) {
    if (i == 5 || i == 0) {
      v16f gi[3];
      gemm_op_bf16<3, 16, 0, false>(rs, PACKED.bt_off[i == 5 ? TOP_SP5_IPE : TOP_SP0], 0, lane, h, pk, gi, nullptr);
      ipe_vjp_accum_lds(gi, X, col, h, gl);
    }
    if (i > 0) {
      gemm_chain_bf16_shared<false>(rs, PACKED.bt_off[i - 1], 0, lane, h, wave, pk, out, ring, NoStepHook());
#pragma unroll
      for (int l = 7; l > 0; --l)
#pragma unroll
        for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
      mask_pack(out, M[7], pk);
    }
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) gl[b] += __shfl_xor(gl[b], 32, 64);
  const float gx[3] = {-gl[2], -gl[1], -gl[0]};
  const float ng = sqrtf(fmaxf((gx[0] * gx[0] + gx[1] * gx[1]) + gx[2] * gx[2], EPS32));
#pragma unroll
  for (int b = 0; b < 3; ++b) nrm_out[b] = -(gx[b] / ng);
}

/* density_normals on the split-f16 chains: the same VJP with 22-bit deltas; d feature / d mean recomputed exactly as in
 * the fp32 kernel (ipe_vjp_accum) */
template <bool SHARED>
__device__ __forceinline__ void density_normals_split(__amdgpu_buffer_rsrc_t rs, int lane, int h, int wave, char *ring, v16f (&out)[8], v4uu (&ph)[16], v4uu (&pl)[16],
                                                      unsigned (&M)[8][4], const float lm[3], const float lv[3], float nrm_out[3]) {
  load_acc<8>(rs, PACKED.wd_off, h, out);
  float c = 1.0f;                                /* per-sample power-of-two factor the chain carries (mask_split) */
  mask_split(out, M[7], ph, pl, c);
  float gl[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int i = 7; i >= 0; --i) {
    if (i == 5 || i == 0) {
      v16f gi[3];
      gemm_op_split<3, 16, 0, false>(rs, PACKED.ht_off[i == 5 ? TOP_SP5_IPE : TOP_SP0], 0, lane, h, ph, pl, gi, nullptr);
      const float inv = 1.0f / c;
#pragma unroll
      for (int blk = 0; blk < 3; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) gi[blk][r] *= inv;
      ipe_vjp_accum(gi, lm, lv, h, gl);
    }
    if (i > 0) {
      if constexpr (SHARED) gemm_chain_split_shared<false>(rs, PACKED.ht_off[i - 1], 0, lane, h, wave, ph, pl, out, ring, NoStepHook());
      else gemm_op_split<8, 16, 0, false>(rs, PACKED.ht_off[i - 1], 0, lane, h, ph, pl, out, nullptr);
#pragma unroll
      for (int l = 7; l > 0; --l)
#pragma unroll
        for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
      mask_split(out, M[7], ph, pl, c);
    }
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) gl[b] += __shfl_xor(gl[b], 32, 64);
  const float gx[3] = {-gl[2], -gl[1], -gl[0]};
  const float ng = sqrtf(fmaxf((gx[0] * gx[0] + gx[1] * gx[1]) + gx[2] * gx[2], EPS32));
#pragma unroll
  for (int b = 0; b < 3; ++b) nrm_out[b] = -(gx[b] / ng);
}

/* ... and for a general IPE basis (see density_normals_gb) */
template <typename Lift>
__device__ __forceinline__ void density_normals_split_gb(__amdgpu_buffer_rsrc_t rs, int lane, int h, v16f (&out)[8], v4uu (&ph)[16], v4uu (&pl)[16],
                                                         unsigned (&M)[8][4], int groups, const float *basis, Lift &&lift, float nrm_out[3]) {
  load_acc<8>(rs, PACKED.wd_off, h, out);
  float c = 1.0f;
  mask_split(out, M[7], ph, pl, c);
  float gw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int i = 7; i >= 0; --i) {
    if (i == 5 || i == 0) {
      const float inv = 1.0f / c;
#pragma unroll 1
      for (int gq = 0; gq < groups; ++gq) {
        v16f gi[3];
        const int a_off = gq == 0 ? PACKED.ht_off[i == 5 ? TOP_SP5_IPE : TOP_SP0] : pext_ht_off(i == 5 ? 1 : 0, gq);
        gemm_op_split<3, 16, 0, false>(rs, a_off, 0, lane, h, ph, pl, gi, nullptr);
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) gi[blk][r] *= inv;
        float lm[3], lv[3], gl[3] = {0.0f, 0.0f, 0.0f};
        lift(gq, lm, lv);
        ipe_vjp_accum(gi, lm, lv, h, gl);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) gw[cc] += gl[b] * basis[9 * gq + 3 * b + cc];
      }
    }
    if (i > 0) {
      gemm_op_split<8, 16, 0, false>(rs, PACKED.ht_off[i - 1], 0, lane, h, ph, pl, out, nullptr);
#pragma unroll
      for (int l = 7; l > 0; --l)
#pragma unroll
        for (int q = 0; q < 4; ++q) M[l][q] = M[l - 1][q];
      mask_split(out, M[7], ph, pl, c);
    }
  }
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) gw[cc] += __shfl_xor(gw[cc], 32, 64);
  const float ng = sqrtf(fmaxf((gw[0] * gw[0] + gw[1] * gw[1]) + gw[2] * gw[2], EPS32));
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) nrm_out[cc] = -(gw[cc] / ng);
}

/* STAGE: MLP.__call__ on caller-supplied Gaussians (no resampling, no compositing): the per-sample
 * outputs of models.py:533-750 for means / covariances given per sample. */
/* BFC (training forward only): the MLP chains on v_mfma_f32_32x32x16_bf16 (cfg.precision = BF16 with cfg.training):
 * activations rounded to bf16 once per layer (what ACT then holds), everything per sample fp32. */
/* SPC (training forward only): the MLP chains on split-f16 operands (cfg.precision = F16X2 with cfg.training): 22-bit
 * products, fp32 everything else, ACT in the fp32 format -- the parity-grade fast training forward. */
/* GB: general IPE basis (cfg.ipe_groups = G > 1 groups of three directions, refnerf_layout.h): the groups pass through
 * the X tile one after the other in layers 0 and 5, each recomputed from the ray where it is consumed (nothing of it
 * stays live across the trunk) */
template <bool TRAIN, bool STAGE = false, bool BFC = false, bool SPC = false, bool GB = false>
__device__ __forceinline__ void level_fwd_f32_body(const LevelArgs &A) {
  static_assert(!BFC || (TRAIN && !STAGE), "bf16 chains: training forward only");
  static_assert(!SPC || (TRAIN && !STAGE && !BFC), "split-f16 chains: training forward only");
  static_assert(!GB || (!BFC && !(STAGE && SPC)), "general IPE basis: the fp32 skeleton with fp32 or split-f16 chains");
  /* split chains on the built-in basis save the layer inputs as hi / lo pair units (REFNERF_ACT_F16X2); a general basis keeps
   * fp32 rows (its tail matrix and tail job table are fp32) */
  constexpr bool PAIRS = SPC && !GB;
  constexpr bool SPLIT_RING = SPC && !GB && REFNERF_SPLIT_SHARED != 0;   /* 256 -> 256 chain layers through the shared weight-stream ring */
  extern __shared__ __attribute__((aligned(16))) float smem[];
  RN_STAMP(A, 0);
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int rpw = A.rpw;
  const int ray0 = blockIdx.x * rpw;
  const int n_tot = rpw * N;                 /* samples owned by this workgroup */

  float *X = smem;                               /* [DIR_PAD][T_TILE]              */
  float *HD = X + DIR_PAD * T_TILE;              /* [HD_ROWS][T_TILE]              */
  float *TD = HD + HD_ROWS * T_TILE;             /* [rpw][N+1] metric distances    */
  float *XP = TD + rpw * (N + 1);                /* [rpw][N+1] CDF knots for the percentiles */
  float *PS = XP + rpw * (N + 1);                /* [n_tot][NPS_TRAIN]             */
  float *PX = PS + n_tot * NPS_TRAIN;            /* [T_TILE][3] grad_pred of the pass */
  float *NRM = PX + 3 * T_TILE;                  /* [rpw] |direction| per ray      */

  if constexpr (!STAGE) {
    resample_phase(A, X, TD, NRM, ray0, wave, lane);    /* P0 */
    __syncthreads();
  }

  /* ---------------- per-pass MLP over 32-sample blocks ---------------- */
  const int col = wave * 32 + sl;                /* this lane's column in X / HD */
  const float *xl = X + h * T_TILE + col;
  v16f in[8], out[8];
  v4uu pk[16];                                   /* packed bf16 layer input (bf16 chains) / hi halves (split chains); dead otherwise */
  v4uu pl[16];                                   /* lo halves (split chains only) */
  const float *xc = X + col;

  RN_STAMP(A, 1);
  for (int pass0 = 0; pass0 < n_tot; pass0 += T_TILE) {
    RN_STAMP(A, 2);
    /* The image descriptor is rebuilt from a laundered pointer in every pass: with a loop-invariant descriptor the
     * compiler hoisted ~100 weight / bias loads of the pass in front of the loop (one pass per workgroup at the usual
     * shapes: nothing gained) and spilled them -- 24 k cycles of spill traffic before the first pass began. */
    const void *packed_l = A.packed;
    long long act_pitch = A.act_pitch;             /* same for the row pitch of ACT: hoisted 64-bit row origins, all spilled */
    asm volatile("" : "+s"(packed_l), "+s"(act_pitch));
    constexpr long long rpitch = RB;               /* unit pitch of the blocked ACT / DELTA rows (refnerf_layout.h) */
    int hdb = DIR_PAD * T_TILE + col;                /* the HD tile (beyond the 64 KB immediate range) through one laundered base */
    asm volatile("" : "+v"(hdb));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)packed_l, 0, (GB ? PACKED_EXT_TOTAL : PACKED.total) * 4, 0x00020000);
    const int g = pass0 + col;                   /* sample index inside the workgroup */
    const int rl = g / N, si = g - rl * N;
    const int ray = ray0 + rl;
    const bool valid = (g < n_tot) && (ray < A.R);
    const int rayc = valid ? ray : (A.R - 1);
    const bool save = TRAIN && !STAGE && A.act != nullptr && valid;   /* keep the layer inputs for the backward */
    const size_t gsx = valid ? (size_t)ray * N + si : 0;
    const size_t rcol = (size_t)rb_col((long long)gsx, act_units(BFC));   /* this sample's column in the blocked ACT rows */
    float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f}, v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if constexpr (!STAGE) {
        o[i] = A.rays.d_origins[(size_t)rayc * 3 + i];
        d[i] = A.rays.d_directions[(size_t)rayc * 3 + i];
      }
      v[i] = A.rays.d_viewdirs[(size_t)rayc * 3 + i];
    }
    /* P1: conical frustum -> lifted Gaussian -> IPE (half 0: sin, half 1: cos) */
    float lm[3], lv[3];
    /* general basis: lifted mean / variance of this lane's sample on the three directions of group gq (recomputed from
     * the ray where it is needed: nothing of it stays live across the trunk) */
    auto lift_group = [&](int gq, float (&gm)[3], float (&gv)[3]) {
      float mean[3], cov[9];
      if constexpr (STAGE) {
        /* the caller's Gaussian: full covariance, or its diagonal (zeros elsewhere) */
        const size_t sidx = valid ? (size_t)ray * N + si : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) mean[i] = A.g_means[sidx * 3 + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) cov[i] = A.cov_full ? A.g_covs[sidx * 9 + i] : ((i % 4 == 0) ? A.g_covs[sidx * 3 + i / 4] : 0.0f);
      } else {
        float og[3], dg[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          og[i] = A.rays.d_origins[(size_t)rayc * 3 + i];
          dg[i] = A.rays.d_directions[(size_t)rayc * 3 + i];
        }
        const float *td = TD + (valid ? rl : 0) * (N + 1);
        cast_sample_full(og, dg, A.rays.d_radii[rayc], td[valid ? si : 0], td[valid ? si + 1 : 1], cfg.ray_shape, mean, cov);
      }
      const float *bs = reinterpret_cast<const float *>(A.packed) + PEXT_BASIS + 9 * gq;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float bb[3] = {bs[3 * b], bs[3 * b + 1], bs[3 * b + 2]};
        lift_onto(mean, cov, bb, gm[b], gv[b]);
        if (cfg.disable_integration) gv[b] = 0.0f;
      }
    };
    /* ... and its IPE features into the X tile; `keep` (training, first use): also into their rows of ACT (group 0) / of
     * the tail matrix behind ACT (groups 1.., refnerf_layout.h: ACT_EXT_*) for the weight-gradient GEMM */
    auto ipe_group = [&](int gq, bool keep) {
      float gm[3], gv[3];
      lift_group(gq, gm, gv);
      float *act_ext = A.act + (size_t)ACT_ALLOC_ROWS * (size_t)act_pitch;
      const size_t xcol = (size_t)rb_col((long long)gsx, ACT_EXT_UNITS);
#pragma unroll 1
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          /* split chains (this kernel is also the F16X2 inference mode of a general basis): the split eval kernel's feature
           * (exact argument, 1.5-ulp reduced sine, hardware exp2: < 2e-7, below the 2^-22 of the operands it feeds) at a third of
           * the libm cost; the octahedron training chains keep libm (their nine-term totals are pinned to 1e-5 of the oracle's) */
          const float fe = SPC ? ipe_feature_split(gm[b], gv[b], j, h) : ipe_feature(gm[b], gv[b], j, h);
          X[(48 * h + j * 3 + b) * T_TILE + col] = fe;
          if constexpr (TRAIN && !STAGE) {
            if (keep && save) {
              if (gq == 0) store_row1(A.act, rpitch, ACT_IPE + 48 * h + j * 3 + b, rcol, fe);
              else store_row1(act_ext, rpitch, (gq - 1) * IPE_DIM + 48 * h + j * 3 + b, xcol, fe);
            }
          }
        }
    };
    if constexpr (GB) {
      lm[0] = lm[1] = lm[2] = 0.0f; lv[0] = lv[1] = lv[2] = 0.0f;
      ipe_group(0, true);
    } else {
      if constexpr (STAGE) {
        /* coord.lift_and_diagonalize (coord.py:129-133) with the octahedron/1 basis:
         * lifted mean = (-z,-y,-x), lifted var = (C_zz, C_yy, C_xx) */
        const size_t sidx = valid ? (size_t)ray * N + si : 0;
        const float *m = A.g_means + sidx * 3;
        lm[0] = -m[2]; lm[1] = -m[1]; lm[2] = -m[0];
        if (A.cov_full) { const float *c = A.g_covs + sidx * 9; lv[0] = c[8]; lv[1] = c[4]; lv[2] = c[0]; }
        else { const float *c = A.g_covs + sidx * 3; lv[0] = c[2]; lv[1] = c[1]; lv[2] = c[0]; }
      } else {
        float radius = A.rays.d_radii[rayc];
        const float *td = TD + (valid ? rl : 0) * (N + 1);
        float t0 = td[valid ? si : 0], t1 = td[valid ? si + 1 : 1];
        cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
        if (cfg.disable_integration) { lv[0] = 0.0f; lv[1] = 0.0f; lv[2] = 0.0f; }        /* models.py:228-231 */
      }
      float fe_prev = 0.0f;
#pragma unroll 1
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const float fe = ipe_feature<BFC>(lm[b], lv[b], j, h);   /* bf16 chains: hardware sin / exp2, as the bf16 eval kernel (the MLP rounds its inputs to 8 bits) */
          X[(48 * h + j * 3 + b) * T_TILE + col] = fe;
          if constexpr (PAIRS) {
            /* features 3 j + b come in row order: every second one completes a pair (rows 48 h + q - 1, 48 h + q) */
            const int q = j * 3 + b;
            if (q & 1) { if (save) store_pair_split(A.act, rpitch, ACT_IPE + 48 * h + q - 1, rcol, fe_prev, fe); }
            else fe_prev = fe;
            continue;
          }
          /* bf16 chains: (e sin, e cos) of every (j, b) once more as a bf16 pair in tile rows 128.. (free until P4): the
           * density-normal VJP needs exactly these as d feature / d mean (ipe_vjp_accum_lds) */
          if constexpr (BFC) reinterpret_cast<unsigned short *>(X)[((BNECK + j * 3 + b) * T_TILE + col) * 2 + h] = (unsigned short)cvt_pk_bf16(fe, fe);
          if constexpr (TRAIN && !STAGE) { if (save) store_row1<BFC>(A.act, rpitch, ACT_IPE + 48 * h + j * 3 + b, rcol, fe); }
        }
    }
    wave_sync();

    RN_STAMP(A, 3);
    /* P2: spatial MLP (models.py:576-580) */
    unsigned M[TRAIN ? 8 : 1][4];                /* ReLU masks of the spatial layers (training) */
    auto act_hook = [&](int row0) {              /* bf16 chains: the packed layer input leaves for ACT, 8 rows per k-step */
      return [&, hk = PairStoreHook(A.act, rpitch, row0, rcol, h, save)](int t) mutable {
#pragma unroll
        for (int e = 0; e < 4; ++e) hk(4 * t + e, pk[t][e]);       /* the k-step's B fragment as it is */
      };
    };
    auto row_hook = [&](int row0) {              /* split chains: the layer input leaves 8 units per k-step: fp32 rows, or */
      return [&, hk = RowStoreHook(A.act, rpitch, row0, rcol, h, save)](int t, int quarter = -1) mutable {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (quarter < 0 || (e >> 1) == quarter) {
            /* (REFNERF_ACT_F16X2) the fragment dwords themselves: unit 2j = hi halves, 2j + 1 = lo halves of rows (2j, 2j + 1) */
            if constexpr (PAIRS) hk.raw(8 * t + e, (e & 1) ? pl[t][e >> 1] : pk[t][e >> 1]);
            else hk(8 * t + e, split_elem(pk[t], pl[t], e));
          }
      };
    };
    /* general basis: groups 1..G-1 of layer L (0: layer 0, 1: layer 5) accumulate into `out` through the same X rows */
    auto more_groups = [&](int L) {
#pragma unroll 1
      for (int gq = 1; gq < cfg.ipe_groups; ++gq) {
        wave_sync();                               /* the previous group's X reads are done */
        ipe_group(gq, L == 0);
        wave_sync();
        if constexpr (SPC) gemm_op_split<8, 0, BF_IPE_STEPS, false, NoStepHook, (1 << 30), true>(rs, pext_hf_off(L, gq), 0, lane, h, pk, pl, out, xc);
        else gemm_op<8, 8, false, false, NoStepHook, rn::PF, true>(rs, pext_fwd_off(L, gq), 0, lane, h, in, out, xl, IPE_DIM / 2);
      }
    };
    if constexpr (SPC) {
      gemm_op_split<8, 0, BF_IPE_STEPS, true>(rs, PACKED.hf_off[0], PACKED.op[0].b_off, lane, h, pk, pl, out, xc);
      if constexpr (GB) more_groups(0);
      relu_mask_split(out, M[7], pk, pl);
    } else if constexpr (BFC) {
      gemm_op_bf16<8, 0, BF_IPE_STEPS, true>(rs, PACKED.bf_off[0], PACKED.op[0].b_off, lane, h, pk, out, xc);
      relu_mask_pack(out, M[7], pk);
    } else {
      gemm_op<8, 8, false>(rs, PACKED.op[0].a_off, PACKED.op[0].b_off, lane, h, in, out, xl, PACKED.op[0].lds_steps);
      if constexpr (GB) more_groups(0);
      if constexpr (TRAIN) relu_mask_into(out, in, M[7]); else relu_into(out, in);
    }
    auto save_mask = [&](int layer, const unsigned (&mk)[4]) {
      if constexpr (BFC) {
        if (save) smb_store(A.act, act_pitch, gsx, h, SMB_MASK + layer, (v4u){mk[0], mk[1], mk[2], mk[3]});
      } else if (save) {
#pragma unroll
        for (int q = 0; q < 4; ++q) store_row1(A.act, rpitch, ACT_MASK + 8 * layer + 4 * h + q, rcol, __builtin_bit_cast(float, mk[q]));
      }
    };
    if constexpr (TRAIN && !STAGE) { if (A.act) save_mask(0, M[7]); }
#pragma unroll 1
    for (int op = 1; op < 8; ++op) {
      /* training: the layer input leaves for the ACT matrix through the store hook (one row per k-step) */
      if constexpr (SPC) {
        if constexpr (GB) { if (op == 5) { wave_sync(); ipe_group(0, false); wave_sync(); } }     /* X holds the last group of layer 0 */
        if (op == 5) gemm_op_split<8, 16, BF_IPE_STEPS, true>(rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, pk, pl, out, xc,
                                                             row_hook(ACT_SP + (op - 1) * WIDTH));
        else if constexpr (SPLIT_RING) gemm_chain_split_shared<true>(rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, wave, pk, pl, out,
                                                                     reinterpret_cast<char *>(smem) + A.ring_off, row_hook(ACT_SP + (op - 1) * WIDTH));
        else gemm_op_split<8, 16, 0, true>(rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, pk, pl, out, xc,
                                           row_hook(ACT_SP + (op - 1) * WIDTH));
        if constexpr (GB) { if (op == 5) more_groups(1); }
      } else if constexpr (BFC) {
        if (op == 5) gemm_op_bf16<8, 16, BF_IPE_STEPS, true>(rs, PACKED.bf_off[op], PACKED.op[op].b_off, lane, h, pk, out, xc,
                                                            act_hook(ACT_SP + (op - 1) * WIDTH));
        else gemm_chain_bf16_shared<true>(rs, PACKED.bf_off[op], PACKED.op[op].b_off, lane, h, wave, pk, out,
                                          reinterpret_cast<char *>(smem) + A.ring_off, act_hook(ACT_SP + (op - 1) * WIDTH));
      } else if constexpr (TRAIN && !STAGE) {
        if constexpr (GB) { if (op == 5) { wave_sync(); ipe_group(0, false); wave_sync(); } }
        gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps,
                            RowStoreHook(A.act, rpitch, ACT_SP + (op - 1) * WIDTH, rcol, h, save));
        if constexpr (GB) { if (op == 5) more_groups(1); }
      } else {
        if constexpr (GB) { if (op == 5) { wave_sync(); ipe_group(0, false); wave_sync(); } }     /* X holds the last group of layer 0 */
        gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps);
        if constexpr (GB) { if (op == 5) more_groups(1); }
      }
      if constexpr (TRAIN) {
#pragma unroll
        for (int l = 0; l < 7; ++l)
#pragma unroll
          for (int q = 0; q < 4; ++q) M[l][q] = M[l + 1][q];
        if constexpr (SPC) relu_mask_split(out, M[7], pk, pl);
        else if constexpr (BFC) relu_mask_pack(out, M[7], pk);
        else relu_mask_into(out, in, M[7]);
        if constexpr (!STAGE) { if (A.act) save_mask(op, M[7]); }
      } else relu_into(out, in);
    }
    RN_STAMP(A, 4);
    /* P3: heads (models.py:582,613,634-645): 4 bottleneck blocks + 1 scalar block */
    {
      v16f hd[5];
      if constexpr (BFC) { if (save) smb_store_pk(A.act, act_pitch, gsx, h, SMB_X7, pk); }
      if constexpr (SPC)
        gemm_op_split<5, 16, 0, true>(rs, PACKED.hf_off[OP_HEADS], PACKED.op[OP_HEADS].b_off, lane, h, pk, pl, hd, xc,
                                      row_hook(ACT_SP + 7 * WIDTH));
      else if constexpr (BFC)
        gemm_op_bf16<5, 16, 0, true>(rs, PACKED.bf_off[OP_HEADS], PACKED.op[OP_HEADS].b_off, lane, h, pk, hd, xc,
                                     act_hook(ACT_SP + 7 * WIDTH));
      else if constexpr (TRAIN && !STAGE)
        gemm_op<5, 8, true>(rs, PACKED.op[OP_HEADS].a_off, PACKED.op[OP_HEADS].b_off, lane, h, in, hd, xl, 0,
                            RowStoreHook(A.act, rpitch, ACT_SP + 7 * WIDTH, rcol, h, save));
      else
        gemm_op<5, 8, true>(rs, PACKED.op[OP_HEADS].a_off, PACKED.op[OP_HEADS].b_off, lane, h, in, hd, xl, 0);
      __builtin_amdgcn_wave_barrier();          /* all IPE reads of this wave are done */
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * T_TILE + col] = hd[blk][r];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < HD_ROWS) X[hdb + row * T_TILE] = hd[4][r];
      }
      if constexpr (PAIRS) { if (A.act) store_rows_split<4>(A.act, rpitch, ACT_DIN, rcol, h, save, hd); }
      else if constexpr (TRAIN && !STAGE) { if (A.act) store_rows<4, BFC>(A.act, rpitch, ACT_DIN, rcol, h, save, hd); }
    }
    wave_sync();

    SampleHeads sh;
    RN_STAMP(A, 5);
    if constexpr (SPC && GB) {
      /* (this kernel also serves cfg.precision = F16X2 in inference for a general basis: no normals then) */
      if (cfg.training) density_normals_split_gb(rs, lane, h, out, pk, pl, M, cfg.ipe_groups, reinterpret_cast<const float *>(A.packed) + PEXT_BASIS, lift_group, sh.normals);
      else { sh.normals[0] = 0.0f; sh.normals[1] = 0.0f; sh.normals[2] = 0.0f; }
    } else if constexpr (SPC) density_normals_split<SPLIT_RING>(rs, lane, h, wave, reinterpret_cast<char *>(smem) + A.ring_off, out, pk, pl, M, lm, lv, sh.normals);
    else if constexpr (BFC) density_normals_bf16(rs, lane, h, wave, reinterpret_cast<char *>(smem) + A.ring_off, out, pk, M, X, col, sh.normals);
    else if constexpr (TRAIN && GB) density_normals_gb(rs, lane, h, in, out, M, xl, cfg.ipe_groups, reinterpret_cast<const float *>(A.packed) + PEXT_BASIS, lift_group, sh.normals);
    else if constexpr (TRAIN) density_normals(rs, lane, h, in, out, M, lm, lv, xl, sh.normals);

    RN_STAMP(A, 6);
    /* P4: activations, reflection, IDE (models.py:611-686) */
    {
      float gp[3], raw_dif[3], raw_tint[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        gp[i] = X[hdb + (1 + i) * T_TILE];
        raw_dif[i] = X[hdb + (5 + i) * T_TILE];
        raw_tint[i] = X[hdb + (8 + i) * T_TILE];
      }
      sample_heads(cfg, X[hdb + 0 * T_TILE], gp, X[hdb + 4 * T_TILE], raw_dif, raw_tint, v, sh);
      const int xhi = tile_hi(col);
      float *xi = X + xhi + IDE_TERMS * h * T_TILE;   /* row BNECK (= 128) + 36 h */
      auto put = [&](int q, float val) {
        xi[q * T_TILE] = val;
        if constexpr (PAIRS) return;             /* pair units: stored from the tile below (the encoders emit out of row order) */
        if constexpr (TRAIN && !STAGE) { if (save) store_row1<BFC>(A.act, rpitch, ACT_DIN + BNECK + IDE_TERMS * h + q, rcol, val); }
      };
      if (cfg.dir_enc == REFNERF_DIRENC_POSENC) posenc_eval(sh.refd[0], sh.refd[1], sh.refd[2], h, put);   /* models.py:487-492 */
      else ide_eval(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, h, put);
      if (h == 0) {
        X[tile_idx(BNECK + IDE_DIM, col, xhi)] = sh.dot;
        if constexpr (TRAIN && !STAGE && !PAIRS) { if (save) store_row1<BFC>(A.act, rpitch, ACT_DIN + BNECK + IDE_DIM, rcol, sh.dot); }
      } else {
#pragma unroll
        for (int q = DIR_IN; q < DIR_PAD; ++q) X[tile_idx(q, col, xhi)] = 0.0f;
      }
      if constexpr (PAIRS) {
        /* this lane's own 36 encoder outputs back from the tile, two rows per pair; half 0 adds (n.v, 0) and the (0, 0) pad pair */
        if (save) {
#pragma unroll 1
          for (int q = 0; q < IDE_TERMS; q += 2)
            store_pair_split(A.act, rpitch, ACT_DIN + BNECK + IDE_TERMS * h + q, rcol, xi[q * T_TILE], xi[(q + 1) * T_TILE]);
          if (h == 0) {
            store_pair_split(A.act, rpitch, ACT_DIN + BNECK + IDE_DIM, rcol, sh.dot, 0.0f);
            store_pair_split(A.act, rpitch, ACT_DIN + BNECK + IDE_DIM + 2, rcol, 0.0f, 0.0f);
          }
        }
      }
    }
    wave_sync();

    RN_STAMP(A, 7);
    /* P5: directional MLP (models.py:690-694) + rgb (699-700) */
    if constexpr (SPC) {
      unsigned mk[4];
      gemm_op_split<8, 0, BF_DIN_STEPS, true, NoStepHook, DIR_PAD - 1>(rs, PACKED.hf_off[9], PACKED.op[9].b_off, lane, h, pk, pl, out, xc);
      relu_mask_split(out, mk, pk, pl);
      if (A.act) save_mask(8, mk);
    } else if constexpr (BFC) {
      unsigned mk[4];
      gemm_op_bf16<8, 0, BF_DIN_STEPS, true, NoStepHook, DIR_PAD - 1>(rs, PACKED.bf_off[9], PACKED.op[9].b_off, lane, h, pk, out, xc);
      relu_mask_pack(out, mk, pk);
      if (A.act) save_mask(8, mk);
    } else {
      gemm_op<8, 8, false>(rs, PACKED.op[9].a_off, PACKED.op[9].b_off, lane, h, in, out, xl, PACKED.op[9].lds_steps);
      if constexpr (TRAIN && !STAGE) {
        unsigned mk[4];
        relu_mask_into(out, in, mk);
        if (A.act) save_mask(8, mk);
      } else relu_into(out, in);
    }
#pragma unroll 1
    for (int op = 10; op < 17; ++op) {
      if constexpr (SPC) {
        if (op == 14) gemm_op_split<8, 16, BF_DIN_STEPS, true, decltype(row_hook(0)), DIR_PAD - 1>(
            rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, pk, pl, out, xc, row_hook(ACT_VD + (op - 10) * WIDTH));
        else if constexpr (SPLIT_RING) gemm_chain_split_shared<true>(rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, wave, pk, pl, out,
                                                                     reinterpret_cast<char *>(smem) + A.ring_off, row_hook(ACT_VD + (op - 10) * WIDTH));
        else gemm_op_split<8, 16, 0, true>(rs, PACKED.hf_off[op], PACKED.op[op].b_off, lane, h, pk, pl, out, xc,
                                           row_hook(ACT_VD + (op - 10) * WIDTH));
      } else if constexpr (BFC) {
        if (op == 14) gemm_op_bf16<8, 16, BF_DIN_STEPS, true, decltype(act_hook(0)), DIR_PAD - 1>(
            rs, PACKED.bf_off[op], PACKED.op[op].b_off, lane, h, pk, out, xc, act_hook(ACT_VD + (op - 10) * WIDTH));
        else gemm_chain_bf16_shared<true>(rs, PACKED.bf_off[op], PACKED.op[op].b_off, lane, h, wave, pk, out,
                                          reinterpret_cast<char *>(smem) + A.ring_off, act_hook(ACT_VD + (op - 10) * WIDTH));
      } else if constexpr (TRAIN && !STAGE)
        gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps,
                            RowStoreHook(A.act, rpitch, ACT_VD + (op - 10) * WIDTH, rcol, h, save));
      else
        gemm_op<8, 8, true>(rs, PACKED.op[op].a_off, PACKED.op[op].b_off, lane, h, in, out, xl, PACKED.op[op].lds_steps);
      if constexpr (TRAIN && !STAGE) {
        unsigned mk[4];
        if constexpr (SPC) relu_mask_split(out, mk, pk, pl);
        else if constexpr (BFC) relu_mask_pack(out, mk, pk);
        else relu_mask_into(out, in, mk);
        if (A.act) save_mask(op - 1, mk);
      } else relu_into(out, in);
    }
    RN_STAMP(A, 8);
    v16f rgbv[1];
    if constexpr (BFC) { if (save) smb_store_pk(A.act, act_pitch, gsx, h, SMB_V7, pk); }
    if constexpr (SPC)
      gemm_op_split<1, 16, 0, true>(rs, PACKED.hf_off[OP_RGB], PACKED.op[OP_RGB].b_off, lane, h, pk, pl, rgbv, xc, row_hook(ACT_VD + 7 * WIDTH));
    else if constexpr (BFC)
      gemm_op_bf16<1, 16, 0, true>(rs, PACKED.bf_off[OP_RGB], PACKED.op[OP_RGB].b_off, lane, h, pk, rgbv, xc, act_hook(ACT_VD + 7 * WIDTH));
    else if constexpr (TRAIN && !STAGE)
      gemm_op<1, 1, true>(rs, PACKED.op[OP_RGB].a_off, PACKED.op[OP_RGB].b_off, lane, h, in, rgbv, xl, 0,
                          RowStoreHook(A.act, rpitch, ACT_VD + 7 * WIDTH, rcol, h, save));
    else
      gemm_op<1, 1, true>(rs, PACKED.op[OP_RGB].a_off, PACKED.op[OP_RGB].b_off, lane, h, in, rgbv, xl, 0);
    /* rows 0..2 live in half 0, regs 0..2; hand them to half 1 as well */
    float raw_rgb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) raw_rgb[i] = __shfl(rgbv[0][i], sl, 64);

    RN_STAMP(A, 9);
    /* P6: colour head (models.py:699-729) */
    if (valid && h == 0) colour_store(A, sh, raw_rgb, PS, PX, n_tot, g, col);
    wave_sync();
    history_flush(A, PS, PX, n_tot, pass0 + wave * 32, wave * 32, (size_t)ray0 * N + pass0 + wave * 32, lane);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  RN_STAMP(A, 10);
  if constexpr (!STAGE) composite_phase(A, TD, XP, PS, n_tot, ray0, wave, lane, X, NRM);   /* P7 */
}

__global__ __launch_bounds__(NTHREADS) void level_fwd_f32(const LevelArgs A) { level_fwd_f32_body<false>(A); }
/* training forward: + density-gradient normals (models.py:603-609) */
__global__ __launch_bounds__(NTHREADS) void level_fwd_train_f32(const LevelArgs A) { level_fwd_f32_body<true>(A); }
/* training forward with the MLP chains on bf16 MFMA (cfg.training && cfg.precision = REFNERF_PREC_BF16) */
__global__ __launch_bounds__(NTHREADS) void level_fwd_train_bf16c(const LevelArgs A) { level_fwd_f32_body<true, false, true>(A); }
/* training forward with the MLP chains on split-f16 operands (cfg.training && cfg.precision = REFNERF_PREC_F16X2) */
__global__ __launch_bounds__(NTHREADS) void level_fwd_train_f16x2c(const LevelArgs A) { level_fwd_f32_body<true, false, false, true>(A); }
/* eval forward with a general IPE basis (cfg.ipe_groups > 1: icosahedron / tesselated bases) */
__global__ __launch_bounds__(NTHREADS) void level_fwd_f32_gb(const LevelArgs A) { level_fwd_f32_body<false, false, false, false, true>(A); }
__global__ __launch_bounds__(NTHREADS) void level_fwd_train_f32_gb(const LevelArgs A) { level_fwd_f32_body<true, false, false, false, true>(A); }
/* ... with the chains on split-f16 operands: the training forward (cfg.training) and, with cfg.training = 0 and no
 * activation buffer, the parity-grade 16-bit inference mode of a general basis */
__global__ __launch_bounds__(NTHREADS) void level_fwd_f16x2c_gb(const LevelArgs A) { level_fwd_f32_body<true, false, false, true, true>(A); }
/* MLP.__call__ stage entry (eval / training) */
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_f32(const LevelArgs A) { level_fwd_f32_body<false, true>(A); }
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_train_f32(const LevelArgs A) { level_fwd_f32_body<true, true>(A); }
/* ... with a general IPE basis */
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_f32_gb(const LevelArgs A) { level_fwd_f32_body<false, true, false, false, true>(A); }
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_train_f32_gb(const LevelArgs A) { level_fwd_f32_body<true, true, false, false, true>(A); }

}  // namespace rn
