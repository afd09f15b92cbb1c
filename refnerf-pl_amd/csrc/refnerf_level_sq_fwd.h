/*
 * refnerf_level_sq_fwd.h -- training forward of one level in the parity-grade 16-bit mode on the EVAL kernel's skeleton
 * (round 5; refnerf_sq_layout.h): level_fwd_split of refnerf_level_bf16.h -- 8 waves, two per SIMD, weights through the
 * LDS-DMA chunk ring, activations in registers, 0 B of scratch -- plus what a training step needs:
 *   * every layer input leaves for the ACT matrix as it is produced: the packed B-fragment dwords themselves (spatial: hi and
 *     lo halves as pair units; directional: the one half the trunk multiplies), behind the MFMAs of the next slice, as
 *     buffer stores on a per-pass window descriptor (no per-lane 64-bit addresses);
 *   * the ReLU sign patterns as lane-local words (8 units per layer and sample);
 *   * the density-gradient normals (internal/models.py:603-609): the VJP of raw_density through the transposed spatial trunk
 *     as a second walk over the same ring (three products, the deltas rescaled per sample BEFORE each contraction by the
 *     layer's column-sum bound G: no pass over the outputs, no overflow for any weights);
 *   * the directional trunk with its W_lo product ([W_hi | W_lo] x: two plain chunks per slice);
 *   * the raw scalar head rows and raw rgb (fp32) for the backward, which then recomputes nothing.
 * The rendezvous of a chunk waits with a COUNTED vmcnt: stores and LDS-DMA share one in-order counter on gfx950, so
 * "vmcnt(K)" with K = the stores issued since the last DMA piece certifies the chunk without draining the stores
 * (__syncthreads would: its fence is s_waitcnt vmcnt(0)).  Rule: a chunk issues its extra vector-memory operations BEFORE its
 * rendezvous step, never behind it.
 * Restates internal/models.py:533-750 (MLP.__call__, training) + render.py:132-254; oracle: rn_level_train.
 */
#pragma once
#include "refnerf_level_bf16.h"
#include "refnerf_sq_layout.h"

namespace rn {

#ifndef REFNERF_SQ_STREAM_AUX
#define REFNERF_SQ_STREAM_AUX 2    /* nt: written once, read by a later kernel */
#endif

/* ---- the pass window of a blocked matrix ([64-sample block][unit][64]): descriptor on the block of the pass's first sample,
 * lane offset = the lane's sample inside it (0xfffffff0: the lane must not store / reads zero) ---- */
struct BlkWin {
  __amdgpu_buffer_rsrc_t rs;
  long long blk0;
};
__device__ __forceinline__ BlkWin blk_window(const void *matrix, long long gs0, int units) {
  BlkWin w;
  w.blk0 = gs0 >> 6;
  const char *base = reinterpret_cast<const char *>(matrix) + (size_t)w.blk0 * (size_t)units * 256u;
  w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x80000000, 0x00020000);
  return w;
}
/* lanes that must not store (and read zero) sit AT num_records: out of range with any lane part or immediate added, no wrap */
constexpr unsigned BLK_NONE = 0x80000000u;
__device__ __forceinline__ unsigned blk_voff(const BlkWin &w, long long gs, int units, bool valid) {
  return valid ? (unsigned)(((gs >> 6) - w.blk0) * (long long)units * 256 + (gs & 63) * 4) : BLK_NONE;
}
/* ... plus `lane_units` units that depend on the lane */
__device__ __forceinline__ unsigned blk_voff_add(unsigned voff, int lane_units) { return voff + (unsigned)lane_units * 256u; }
/* unit `sunit` + `iunit`: sunit goes into the scalar offset (a wave-uniform run-time value: one s_add per store, nothing to
 * hoist), iunit (0..15) into the instruction's 12-bit immediate */
/* (`voff` is laundered at every use: left visible, the loop-invariant sums voff + iunit * 256 are hoisted in front of the layer
 * loops -- one VGPR per distinct immediate, 8 to 16 per window, live across every trunk -- and the instruction selector, which
 * works block by block, then finds no add to fold into the immediate field) */
__device__ __forceinline__ void win_store(const BlkWin &w, unsigned voff, int sunit, int iunit, unsigned dword) {
#ifndef REFNERF_EXPERIMENT_NO_STREAM
  asm volatile("" : "+v"(voff));
#ifdef REFNERF_EXPERIMENT_STORE_LOCAL   /* timing experiment only: every store of a lane lands on the same line (no HBM traffic) */
  sunit &= REFNERF_EXPERIMENT_STORE_LOCAL;
#endif
  __builtin_amdgcn_raw_buffer_store_b32(dword, w.rs, voff + (unsigned)iunit * 256u, sunit * 256, REFNERF_SQ_STREAM_AUX);
#endif
}
__device__ __forceinline__ unsigned win_load(const BlkWin &w, unsigned voff, int sunit, int iunit) {
  asm volatile("" : "+v"(voff));
  return __builtin_amdgcn_raw_buffer_load_b32(w.rs, voff + (unsigned)iunit * 256u, sunit * 256, 0);
}
/* the cycle stamps of the training kernels take their lane test from a lane index formed at the stamp (and only when profiling
 * is on): the kernel's entry value of `lane` need not survive to them */
#undef RN_STAMPW
#define RN_STAMPW(A, slot) do { asm volatile("; RNMARK " #slot); if ((A).prof && blockIdx.x == (gridDim.x >> 1) && fresh_lane() == 0) (A).prof[wave * 32 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
/* a unit index the compiler must treat as a run-time scalar */
__device__ __forceinline__ int opaque_s(int x) {
  asm volatile("" : "+s"(x));
  return x;
}

/* the rendezvous: chunk c + 1 is complete for every wave (every wave's DMA pieces have landed: they were issued before the
 * VMK newest vector-memory operations of this wave), chunk c - 1's slot is free */
template <int VMK>
__device__ __forceinline__ void tq_rendezvous(Pipe &p) {
#ifdef REFNERF_PROF_WAITS   /* cycles in the counted wait / in the barrier (debug builds) */
  const long long t0 = (long long)__builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMK) : "memory");
  const long long t1 = (long long)__builtin_readcyclecounter();
  asm volatile("s_barrier" ::: "memory");
  const long long t2 = (long long)__builtin_readcyclecounter();
  p.t_vm += t1 - t0;
  p.t_bar += t2 - t1;
#elif defined(REFNERF_EXPERIMENT_TQ_SYNC)   /* timing experiment: the eval kernels' rendezvous (fence + barrier: drains stores and LDS reads) */
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#elif defined(REFNERF_EXPERIMENT_TQ_LGKM)   /* timing experiment: counted vmcnt, but the LDS reads drained as __syncthreads does */
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VMK) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VMK) : "memory");
#endif
}

/* stream position of the training forward: [run section] x 2 + [directional section]; of the backward: linear */
template <bool BWD>
__device__ __forceinline__ void tq_issue(Pipe &p, int slot_off, int piece = -1) {
  if (p.dma_left > 0) {
    /* (the wave index is laundered: its comparisons are then formed here -- one s_cmp each -- instead of being hoisted to the
     *  kernel's entry as lane masks, which end up as VGPR booleans in scratch once the SGPRs run out) */
    int wv = p.wave;
    asm volatile("" : "+s"(wv));
    if (wv < 6) {
      lptr_t dst = (lptr_t)(p.wbuf + slot_off + wv * 3072);
      if (piece < 0 || piece == 0) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 0, REFNERF_DMA_AUX);
      if (piece < 0 || piece == 1) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 1024, REFNERF_DMA_AUX);
      if ((piece < 0 || piece == 2) && wv < 5) __builtin_amdgcn_global_load_lds((gptr_t)p.src, dst, 16, 2048, REFNERF_DMA_AUX);
    }
    if (piece >= 0 && piece < 2) return;
    p.src += BF_CHUNK_BYTES;
    p.seq += 1;
    if constexpr (BWD) {
      if (p.seq == TR_BWD) { p.src -= (size_t)TR_BWD * BF_CHUNK_BYTES; p.seq = 0; }
    } else {
      if (p.seq == TR_RUN) p.src -= (size_t)TR_RUN * BF_CHUNK_BYTES;
      else if (p.seq == TR_FWD_PASS) { p.src -= (size_t)TR_FWD * BF_CHUNK_BYTES; p.seq = 0; }
    }
    p.dma_left -= 1;
  }
}
__device__ __forceinline__ void tq_rotate(Pipe &p) {
  const int t = p.cur_off;
  p.cur_off = p.nxt_off;
  p.nxt_off = p.fil_off;
  p.fil_off = t;
}
/* (round 6) the forward's extra workgroup barrier between the second run's VJP and the directional phase (see P4): an idle wave
 * joins it at the same place in the barrier sequence, behind the 2 TR_RUN rendezvous of the two runs */
__device__ __forceinline__ void tq_phase_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool BWD>
__device__ __forceinline__ void tq_idle_pass(Pipe &p) {
#pragma unroll 1
  for (int c = 0; c < (BWD ? TR_BWD : TR_FWD_PASS); ++c) {
    if (!BWD && c == 2 * TR_RUN) tq_phase_barrier();
    tq_rendezvous<0>(p);
    tq_issue<BWD>(p, p.fil_off);
    tq_rotate(p);
  }
}

/* One chunk of the 16x16x32 sections: sq_chunk of refnerf_level_bf16.h with the counted rendezvous.  VMK = vector-memory
 * operations this wave issues between the DMA pieces of the PREVIOUS chunk and this chunk's rendezvous step. */
template <int KIND, bool FIRST, bool PRE, int VMK, typename Hook = NoHook>
__device__ __forceinline__ void tq_chunk(Pipe &p, sq_v8 (&fr)[SQ_NF], const v4uu (&in)[16], SqAcc &acc, SqAcc &nacc, Hook &&hook = Hook()) {
  constexpr int NM = sq_nm<KIND>(), NP = sq_np<KIND>();
  constexpr int RDV = NM / 2 - 1;
  const char *w = p.wbuf + p.cur_off;
  const char *cur = w + 1024 + p.lane * 16;
  const char *nxt = p.wbuf + p.nxt_off + 1024 + p.lane * 16;
  sq_v8 xb[2];
  if (KIND == SQ_X) { xb[0] = lds_frag<MmF16>(p.xps); xb[1] = lds_frag<MmF16>(p.xps + (BT / 2) * 16); }
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
    if (j <= RDV) hook(j);                       /* (the rule: nothing of the caller's behind the rendezvous step) */
    if (KIND == SQ_X) {
      if ((j % 6) == 3 && sl + 1 < 3) xb[0] = lds_frag<MmF16>(p.xps + (sl + 1) * (4 * BT * 16));
      if ((j % 6) == 5 && sl + 1 < 3) xb[1] = lds_frag<MmF16>(p.xps + (sl + 1) * (4 * BT * 16) + (BT / 2) * 16);
    }
    if (j == RDV) {
      tq_rendezvous<VMK>(p);
      tq_issue<false>(p, p.fil_off, 0);
      if (PRE) {
        const v4f *bp = reinterpret_cast<const v4f *>(p.wbuf + p.nxt_off + (p.lane >> 4) * 16);
        nacc.t0 = bp[0];
        nacc.t1 = bp[4];
      }
    }
    if (j == RDV + 4) tq_issue<false>(p, p.fil_off, 1);
    if (j == RDV + 8) tq_issue<false>(p, p.fil_off, 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  tq_rotate(p);
}
/* (a chunk of a section without biases: the transposed ops) */
__device__ __forceinline__ void tq_zero(SqAcc &a) { a.t0 = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; a.t1 = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; }

/* One plain chunk (32x32x16, 32 samples per wave): bf_chunk of refnerf_level_bf16.h with the counted rendezvous. */
template <bool BWD, int KIND, int REAL_L, bool FIRST, int VMK>
__device__ __forceinline__ void tq_bf_chunk(Pipe &p, MmF16::v8 (&a)[AF], const v4uu (&in)[16], const v4uu (&bn)[8], v16f &acc) {
  typedef MmF16 MM;
  typedef MM::v8 v8mm;
  constexpr int KS = (KIND == BF_LDS8) ? 8 : 16;
  constexpr int L0 = (KIND == BF_BNLDS) ? 8 : 0;
  constexpr int RDV = KS / 2 - 1;
  const char *w = p.wbuf + p.cur_off;
  const char *cur = w + 1024 + p.lane * 16;
  const char *nxt = p.wbuf + p.nxt_off + 1024 + p.lane * 16;
  v8mm xr[2];
  /* (REAL_L = 1: the tile holds ONE k-step -- the pad steps, whose weights are zero, re-read it instead of running past it) */
  auto lds_bq = [&](int kl) { return REAL_L == 1 ? lds_frag<MM>(p.xp) : lds_b<MM, REAL_L>(p, kl); };
  if (KIND == BF_LDS8) { xr[0] = lds_bq(0); xr[1] = lds_bq(1); }
  if (FIRST) acc = bias16(w, p.h);
#ifdef REFNERF_EXPERIMENT_ACC2
  v16f acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
#endif
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    v8mm b;
    const bool lds_step = (KIND == BF_LDS8) || (KIND == BF_BNLDS && k >= 8);
    if (lds_step) b = xr[(k - L0) & 1];
    else if (KIND == BF_REG) b = __builtin_bit_cast(v8mm, in[k]);
    else b = __builtin_bit_cast(v8mm, bn[k & 7]);
#ifdef REFNERF_EXPERIMENT_ACC2
    if (k & 1) acc2 = MM::mfma(a[k % AF], b, acc2); else
#endif
    acc = MM::mfma(a[k % AF], b, acc);
    a[k % AF] = (k + AF < KS) ? lds_frag<MM>(cur + (k + AF) * 1024) : lds_frag<MM>(nxt + (k + AF - KS) * 1024);
    if (KIND == BF_LDS8 || KIND == BF_BNLDS) {
      const int kl2 = k + 2 - L0;
      if (kl2 >= 0 && kl2 < 8 && !(KIND == BF_LDS8 && kl2 < 2)) xr[kl2 & 1] = lds_bq(kl2);
    }
    if (k == RDV) {
      tq_rendezvous<VMK>(p);
      tq_issue<BWD>(p, p.fil_off, 0);
    }
    if (k == RDV + 2) tq_issue<BWD>(p, p.fil_off, 1);
    if (k == RDV + 4) tq_issue<BWD>(p, p.fil_off, 2);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef REFNERF_EXPERIMENT_ACC2
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
#endif
  tq_rotate(p);
}

/* x where bit `bit` of `mk` is set, else +0: v_bfe_i32 (0 / all ones) + v_and (keep_if_bit of refnerf_level_f32.h) */
__device__ __forceinline__ float tq_keep(float x, unsigned mk, int bit) {
  const int m = __builtin_amdgcn_sbfe((int)mk, bit, 1);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (unsigned)m);
}
/* NaN-propagating ReLU of the split kernels + its sign bit (refnerf_level_f32.h: relu_bit<true>): bit `bit` of mk */
__device__ __forceinline__ float tq_relu_bit(float y, unsigned &mk, int bit) {
  const float x = (y < 0.0f) ? 0.0f : y;
  const unsigned u = __builtin_bit_cast(unsigned, x);
  mk |= (u < 1u ? u : 1u) << bit;
  return x;
}

/* piece q of a spatial slice's epilogue in the training forward: ReLU, sign bits (bits `bit0` + 2 q, + 1 and + 4 for the T1
 * values), hi / lo split, and the two dwords to their pair units */
template <bool ACT_LO>
__device__ __forceinline__ void tq_epi_piece(const SqAcc &a, int q, v4uu &oh, v4uu &ol, unsigned &mk, int bit0, const BlkWin &aw, unsigned voff, int unit) {
  const float y0 = q == 0 ? a.t0[0] : (q == 1 ? a.t0[2] : (q == 2 ? a.t1[0] : a.t1[2]));
  const float y1 = q == 0 ? a.t0[1] : (q == 1 ? a.t0[3] : (q == 2 ? a.t1[1] : a.t1[3]));
  const int bit = bit0 + 4 * (q >> 1) + 2 * (q & 1);
  const float x0 = tq_relu_bit(y0, mk, bit), x1 = tq_relu_bit(y1, mk, bit + 1);
  unsigned hi, lo;
  split_pair_f16(x0, x1, hi, lo);
  oh[q] = hi;
  ol[q] = lo;
  /* rows 32 s + 16 (q / 2) + 4 b + 2 (q % 2), + 1: the pair's hi unit and its lo unit (4 b rows ride in voff) */
  win_store(aw, voff, unit + 16 * (q >> 1), 2 * (q & 1), hi);
  if (ACT_LO) win_store(aw, voff, unit + 16 * (q >> 1), 2 * (q & 1) + 1, lo);    /* (REFNERF_WGRAD_F16: the weight-gradient GEMM reads the hi halves only) */
}

/* One spatial layer of the training forward: sq_layer + the ACT / mask stores of its OUTPUT (= the next layer's input).
 * `unit` = first unit of that input's pair units; VMK0 = vector-memory operations in front of the layer's first chunk. */
template <bool LAYER0, int VMK0, bool ACT_LO>
__device__ __forceinline__ void tq_layer(Pipe &p, sq_v8 (&fr)[SQ_NF], SqAcc (&accs)[2], bool skip, const v4uu (&in)[16], v4uu (&out)[16],
                                         const BlkWin &aw, unsigned voff_sp, unsigned voff_mk, int unit, int mask_unit) {
  unsigned mk0 = 0u, mk1 = 0u;
  unit = opaque_s(unit);
  mask_unit = opaque_s(mask_unit);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    SqAcc &acc = accs[(ob + 1) & 1];
    SqAcc &prev = accs[ob & 1];
    auto hook = [&](int j) {
      if (ob == 0 || j >= 8 || (j & 1)) return;
      tq_epi_piece<ACT_LO>(prev, j >> 1, out[2 * ob - 2], out[2 * ob - 1], (ob - 1) < 4 ? mk0 : mk1, 8 * ((ob - 1) & 3), aw, voff_sp, unit + 32 * (ob - 1));
    };
    constexpr int VM = ACT_LO ? 8 : 4;
    if constexpr (LAYER0) {
      if (ob == 0) tq_chunk<SQ_X, true, true, VMK0>(p, fr, in, acc, prev, hook);
      else tq_chunk<SQ_X, true, true, VM>(p, fr, in, acc, prev, hook);
    } else {
      if (ob == 0) tq_chunk<SQ_A, true, false, VMK0>(p, fr, in, acc, prev, hook);
      else tq_chunk<SQ_A, true, false, VM>(p, fr, in, acc, prev, hook);
      tq_chunk<SQ_B, false, true, 0>(p, fr, in, acc, prev);
      if (skip) tq_chunk<SQ_X, false, true, 0>(p, fr, in, acc, prev);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) tq_epi_piece<ACT_LO>(accs[0], q, out[14], out[15], mk1, 24, aw, voff_sp, unit + 32 * 7);
  win_store(aw, voff_mk, mask_unit, 0, mk0);
  win_store(aw, voff_mk, mask_unit, 1, mk1);
  __builtin_amdgcn_sched_barrier(0);
}
/* 8 (4: hi halves only) tail stores + 2 mask words in front of the next layer's first chunk */
template <bool ACT_LO> constexpr int tq_vm_layer() { return (ACT_LO ? 8 : 4) + 2; }

/* ---- the density-gradient VJP (16-sample tiles, three products) ---- */
/* power of two that keeps |W^T delta| below 2^15 for a sample whose largest |delta| is m: m rs G < 2^15 */
__device__ __forceinline__ float tq_bound_scale(float m, float G) {
  const float x = m * G;
  int e = 15 - __builtin_amdgcn_frexp_expf(x);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return (x > 0.0f) ? __builtin_ldexpf(1.0f, e) : 1.0f;
}
/* value `v` of lane `src` (ds_bpermute on an index formed here: HIP's __shfl / __shfl_xor take this lane's index from the
 * kernel's entry, where it is computed once and then rides -- in scratch -- to every shuffle of the kernel) */
__device__ __forceinline__ float tq_lane_read(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}
/* largest value over the four lanes (b = 0..3) that hold one sample */
__device__ __forceinline__ float tq_max4(float m) {
  const int ln = fresh_lane();
  m = fmaxf(m, tq_lane_read(m, ln ^ 16));
  return fmaxf(m, tq_lane_read(m, ln ^ 32));
}
__device__ __forceinline__ float tq_sum4(float s) {
  const int ln = fresh_lane();
  s += tq_lane_read(s, ln ^ 16);
  return s + tq_lane_read(s, ln ^ 32);
}
/* piece q of a transposed slice's epilogue: the recorded sign bits, the sample's factor, running max, hi / lo split */
__device__ __forceinline__ void tq_vjp_piece(const SqAcc &a, int q, v4uu &oh, v4uu &ol, unsigned mk, int bit0, float rs, float &mx) {
  const float y0 = q == 0 ? a.t0[0] : (q == 1 ? a.t0[2] : (q == 2 ? a.t1[0] : a.t1[2]));
  const float y1 = q == 0 ? a.t0[1] : (q == 1 ? a.t0[3] : (q == 2 ? a.t1[1] : a.t1[3]));
  const int bit = bit0 + 4 * (q >> 1) + 2 * (q & 1);
  const float x0 = tq_keep(y0, mk, bit) * rs, x1 = tq_keep(y1, mk, bit + 1) * rs;
  mx = fmaxf(mx, fmaxf(fabsf(x0), fabsf(x1)));
  asm volatile("" : "+v"(mx));                   /* (pinned: the optimiser otherwise gathers a layer's maximum into one tree behind its last slice) */
  unsigned hi, lo;
  split_pair_f16(x0, x1, hi, lo);
  oh[q] = hi;
  ol[q] = lo;
}
/* One transposed 256 -> 256 layer: out = mask (W^T in) rs.  The sign words of the layer BELOW (whose output the result is
 * the gradient of: units mask_unit, + 1) are fetched behind the first MFMA -- two vector-memory operations in front of the
 * first rendezvous, needed two chunks later; `mx` returns the largest |out| of this lane. */
__device__ __forceinline__ void tq_vjp_layer(Pipe &p, sq_v8 (&fr)[SQ_NF], const v4uu (&in)[16], v4uu (&out)[16], const BlkWin &aw, unsigned voff_mk, int mask_unit, float rs, float &mx) {
  SqAcc accs[2];
  unsigned mk0 = 0u, mk1 = 0u;
  mx = 0.0f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    SqAcc &acc = accs[(ob + 1) & 1];
    SqAcc &prev = accs[ob & 1];
    tq_zero(acc);
    auto hook = [&](int j) {
      if (ob == 0) {
        if (j == 0) { mk0 = win_load(aw, voff_mk, mask_unit, 0); mk1 = win_load(aw, voff_mk, mask_unit, 1); }
        return;
      }
      if (j >= 8 || (j & 1)) return;
      tq_vjp_piece(prev, j >> 1, out[2 * ob - 2], out[2 * ob - 1], (ob - 1) < 4 ? mk0 : mk1, 8 * ((ob - 1) & 3), rs, mx);
    };
    if (ob == 0) tq_chunk<SQ_A, true, false, 2>(p, fr, in, acc, prev, hook);
    else tq_chunk<SQ_A, true, false, 0>(p, fr, in, acc, prev, hook);
    tq_chunk<SQ_B, false, false, 0>(p, fr, in, acc, prev);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) tq_vjp_piece(accs[0], q, out[14], out[15], mk1, 24, rs, mx);
  __builtin_amdgcn_sched_barrier(0);
}
/* d feature / d lifted mean of IPE feature k' = 48 hb + 3 j + axis (coord.py:119-126 differentiated, the variance detached as
 * the rest of the sample geometry): e cos(arg) 2^j with the forward's own argument (ipe_feature_split) */
__device__ __forceinline__ float tq_ipe_dmean(float lm, float lv, int j, int hb) {
  const float sc = __builtin_ldexpf(1.0f, j), sc2 = __builtin_ldexpf(1.0f, 2 * j);
  float x = lm * sc;
  if (hb) x = x + HALF_PI_F;
  const float e = __builtin_amdgcn_exp2f((-0.5f * LOG2E_F) * (lv * sc2));
  return (e * cos_reduced(safe_arg_exact(x))) * sc;
}
/* The 96 IPE rows of a transposed layer (3 slices).  This lane's 24 rows k' = K0 + 4 b (K0 compile-time: 32 ob + 16 (e / 4) +
 * e % 4) are features of axis (K0 + b) mod 3; their derivatives d feature / d mean were parked by the caller in the lane's own
 * columns of the (dead) IPE planes: value v = 8 ob + e at `dfac` + (v / 2) * 4 KB + (v % 2) * 2 KB.  gr[K0 mod 3] += g * that:
 * one LDS read and two VALU per row behind the MFMAs (computed in place, the compiler hoisted all 24 evaluations -- float64
 * range reductions -- in front of the chunks and kept them across five layers: 1 KB / lane of scratch). */
__device__ __forceinline__ void tq_vjp_ipe(Pipe &p, sq_v8 (&fr)[SQ_NF], const v4uu (&in)[16], const char *dfac, float inv, float (&ga)[3]) {
  SqAcc accs[2];
  float gr[3] = {0.0f, 0.0f, 0.0f};
  auto one = [&](const SqAcc &a, int ob, int e) {
    const float g = e < 4 ? a.t0[e & 3] : a.t1[e & 3];
    const int K0 = 32 * ob + 16 * (e >> 2) + (e & 3), v = 8 * ob + e;
    gr[(K0 % 48) % 3] += g * *reinterpret_cast<const float *>(dfac + (v >> 1) * (BT * 16) + (v & 1) * ((BT / 2) * 16));
  };
#pragma unroll
  for (int ob = 0; ob < 3; ++ob) {
    SqAcc &acc = accs[(ob + 1) & 1];
    SqAcc &prev = accs[ob & 1];
    tq_zero(acc);
    /* the previous slice's eight values ride behind this slice's MFMAs, one per hook call */
    auto hook = [&](int j) {
      if (ob == 0 || j >= 8) return;
      one(prev, ob - 1, j);
    };
    tq_chunk<SQ_A, true, false, 0>(p, fr, in, acc, prev, hook);
    tq_chunk<SQ_B, false, false, 0>(p, fr, in, acc, prev);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) one(accs[1], 2, e);
#pragma unroll
  for (int a = 0; a < 3; ++a) ga[a] += gr[a] * inv;
  __builtin_amdgcn_sched_barrier(0);
}

/* ---- directional trunk: [W_hi | W_lo] x, ReLU sign words, the one-half layer inputs to ACT ---- */
/* acc (one 32x32 fp32 tile) -> two packed f16 B fragments of the next layer, ReLU on the packed pairs (pack_pair of
 * refnerf_level_bf16.h), + 16 sign bits at `bit0` of mk, taken from the packed halves (non-zero <=> active): VALU only, no
 * compare results in SGPR pairs (sixteen of those in flight per slice spilled 300 B / lane) */
__device__ __forceinline__ void tq_pair_bits(unsigned w, unsigned &mk, int bit) {
  const unsigned lo = w & 0xffffu, hi = w >> 16;
  mk |= (lo < 1u ? lo : 1u) << bit;
  mk |= (hi < 1u ? hi : 1u) << (bit + 1);
}
__device__ __forceinline__ void tq_pack_acc(const v16f &a, v4uu &f0, v4uu &f1, unsigned &mk, int bit0) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f0[e] = pack_pair<MmF16, true>(a[2 * e], a[2 * e + 1]);
    f1[e] = pack_pair<MmF16, true>(a[8 + 2 * e], a[8 + 2 * e + 1]);
    tq_pair_bits(f0[e], mk, bit0 + 2 * e);
    tq_pair_bits(f1[e], mk, bit0 + 8 + 2 * e);
  }
}
/* the eight dwords of k-steps 2 ob, 2 ob + 1 to their one-half pair rows: unit0 + 16 ob + 8 (t & 1) + 4 (q >> 1) + (q & 1) (+ 2 h in voff) */
__device__ __forceinline__ void tq_store_frag_pair(const BlkWin &aw, unsigned voff, int unit, const v4uu &f0, const v4uu &f1) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    win_store(aw, voff, unit, 4 * (q >> 1) + (q & 1), f0[q]);
    win_store(aw, voff, unit, 8 + 4 * (q >> 1) + (q & 1), f1[q]);
  }
}
template <int KIND0, int REAL0, int VMK0>
__device__ __forceinline__ void tq_dir_layer(Pipe &p, MmF16::v8 (&a)[AF], int second, const v4uu (&in)[16], const v4uu (&bn)[8], v4uu (&out)[16],
                                             const BlkWin &aw, unsigned voff_h2, unsigned voff_h4, int unit, int mask_unit) {
  unsigned mk[4] = {0u, 0u, 0u, 0u};
  unit = opaque_s(unit);
  mask_unit = opaque_s(mask_unit);
  /* the skip layer's bottleneck fragments come back from ACT HERE (32 dwords per lane, L2 hits, once per pass): held from the
   * first directional layer on they are 32 registers live across three layers, eight of them in scratch around the layer loop.
   * (Loads in front of the layer's first rendezvous only make its counted vmcnt wait stricter.) */
  v4uu bnl[8];
  if constexpr (KIND0 == BF_REG) {
    if (second == 2) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) bnl[t][q] = win_load(aw, voff_h2, opaque_s(AQ_DIN + 16 * (t >> 1)), 8 * (t & 1) + 4 * (q >> 1) + (q & 1));
    }
  }
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    v16f acc;
    if (ob == 0) tq_bf_chunk<false, KIND0, REAL0, true, VMK0>(p, a, in, bn, acc);
    else tq_bf_chunk<false, KIND0, REAL0, true, 8>(p, a, in, bn, acc);
    tq_bf_chunk<false, KIND0, REAL0, false, 0>(p, a, in, bn, acc);
    if constexpr (KIND0 == BF_REG) {
      if (second == 2) {
        tq_bf_chunk<false, BF_BNLDS, BF_DIR_REAL_KS, false, 0>(p, a, in, bnl, acc);
        tq_bf_chunk<false, BF_BNLDS, BF_DIR_REAL_KS, false, 0>(p, a, in, bnl, acc);
      }
    }
    tq_pack_acc(acc, out[2 * ob], out[2 * ob + 1], mk[ob >> 1], 16 * (ob & 1));
    tq_store_frag_pair(aw, voff_h2, unit + 16 * ob, out[2 * ob], out[2 * ob + 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) win_store(aw, voff_h4, mask_unit, q, mk[q]);
}
constexpr int TQ_VM_DIR_LAYER = 12;  /* 8 fragment dwords + 4 mask words in front of the next layer's first chunk */

/* ACT_LO: the lo halves of the spatial layer inputs (and of the IPE rows) are written to ACT (cfg.wgrad_mode != REFNERF_WGRAD_F16) */
template <bool ACT_LO>
__device__ __forceinline__ void level_fwd_train_sq_body(const LevelArgs &A) {
  typedef MmF16 MM;
  typedef MM::v8 v8mm;
  typedef MM::t mm_t;
  constexpr int NP = NPS_TRAIN;
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
  float *PS = XP + rpw * (N + 1);                            /* [n_tot][NP]              */
  float *PX = PS + n_tot * NP;                               /* [BT][3] grad_pred of the pass */
  float *NRM = PX + 3 * BT;                                  /* [8] |direction| per ray  */
  const float *RY = NRM + 8;                                 /* [rpw][12] o, d, viewdir, radius per ray */

  const int h = lane >> 5, n = lane & 31;
  const int col = wave * 32 + n;                             /* this lane's sample column (directional phase) */
  const float *KC = reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.packed) + TR_CONST_OFF);

  Pipe p;
  p.src = reinterpret_cast<const char *>(A.packed) + wave * 3072 + lane * 16;
  p.src_end = nullptr;
  p.wbuf = WB;
  p.xp = Xb + (h * BT + col) * 16;
  p.xps = Xb + ((lane >> 4) * BT + wave * 16 + (lane & 15)) * 16;
  p.seq = 0;
  p.cur_off = 0; p.nxt_off = BF_CHUNK_BYTES; p.fil_off = 2 * BF_CHUNK_BYTES;
  p.dma_left = n_pass * TR_FWD_PASS;
  p.lane = lane; p.wave = wave; p.h = h;
  p.t_vm = 0; p.t_bar = 0;
  RN_STAMPW(A, 0);
  tq_issue<false>(p, p.cur_off);                             /* overlaps with the resampler */
  tq_issue<false>(p, p.nxt_off);

  resample_phase<BF_NW, true>(A, reinterpret_cast<float *>(Xb), TD, NRM, ray0, wave, lane);   /* P0: bit-exact CDF */
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
      const float dx = tq_lane_read(val, 3), dy = tq_lane_read(val, 4), dz = tq_lane_read(val, 5);
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
  sq_v8 ar[SQ_NF];
#pragma unroll
  for (int d = 0; d < SQ_NF; ++d) ar[d] = lds_frag<MM>(WB + 1024 + lane * 16 + d * 1024);

  for (int pass0 = 0; pass0 < n_tot; pass0 += BT) {
    /* (round 6) lane constants are formed from a lane index taken where they are needed (fresh_lane: not hoistable, not merged
     * with the kernel's entry value) */
    auto locate = [&](int &g, int &rl, bool &valid) {
      int col_l = wave * 32 + (fresh_lane() & 31), pass_l = pass0;
      asm volatile("" : "+v"(col_l), "+s"(pass_l));
      g = pass_l + col_l;
      rl = g / N;
      valid = (g < n_tot) && (ray0 + rl < A.R);
    };
    /* the ring's upper entries do not cross a pass boundary (every run re-fetches them; the directional chunks keep AF fragments
     * ahead): defined here on every path, or the last spatial chunk's stale pieces ride through the directional trunk in scratch */
#pragma unroll
    for (int d = AF; d < SQ_NF; ++d) asm volatile("" : "=v"(ar[d]));
    {
      const int g0 = pass0 + wave * 32;
      if (g0 >= n_tot || ray0 + g0 / N >= A.R) { tq_idle_pass<false>(p); continue; }
    }
    /* this pass's window of ACT; lane offsets are formed where they are used (from laundered inputs: nothing carried) */
    const long long gs_pass = (long long)ray0 * N + pass0;
    void *act_l = A.act;
    asm volatile("" : "+s"(act_l));
    const BlkWin aw = blk_window(act_l, gs_pass, AQ_UNITS);
    auto load_heads = [&](SampleHeads &sh) {
      int g, rl; bool valid;
      locate(g, rl, valid);
      int ci = wave * 32 + (fresh_lane() & 31), ro = (valid ? rl : 0) * 12;
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

#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { R0[e] = (v4uu){0, 0, 0, 0}; R1[e] = (v4uu){0, 0, 0, 0}; }
      /* the lane's sample of this run: 16 * phase + (lane & 15) of the wave's 32; b = lane >> 4 = its k-group.  Formed per
       * section of the run (IPE | trunk + heads | VJP) from a fresh lane index and laundered loop counters */
      struct RunConsts { int i16, bq, cs, gs; bool vs; unsigned voff_c; };
      auto run_consts = [&]() {
        RunConsts c;
        int ps = pass0, ph = phase;
        asm volatile("" : "+s"(ps), "+s"(ph));
        const int ln = fresh_lane();
        c.i16 = ln & 15; c.bq = ln >> 4;
        c.cs = wave * 32 + 16 * ph + c.i16;                 /* pass column */
        c.gs = ps + c.cs;
        const int rls = c.gs / N;
        c.vs = (c.gs < n_tot) && (ray0 + rls < A.R);
        c.voff_c = blk_voff(aw, (long long)ray0 * N + ps + c.cs, AQ_UNITS, c.vs);     /* (invalid: stays out of range with any lane part added) */
        return c;
      };
      /* lifted mean / variance of this lane's sample (recomputed by the VJP: nothing of it lives across the trunk) */
      auto lift = [&](float (&lm)[3], float (&lv)[3]) {
        int ln = fresh_lane(), ps = pass0, ph = phase;
        asm volatile("" : "+s"(ps), "+s"(ph));
        const int g2 = ps + wave * 32 + 16 * ph + (ln & 15);
        const int r2 = g2 / N, s2 = g2 - r2 * N;
        const bool v2 = (g2 < n_tot) && (ray0 + r2 < A.R);
        float o[3], d[3];
        const float *ry = RY + (v2 ? r2 : 0) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] = ry[i]; d[i] = ry[3 + i]; }
        const float radius = ry[9];
        const float *td = TD + (v2 ? r2 : 0) * (N + 1);
        const float t0 = td[v2 ? s2 : 0], t1 = td[v2 ? s2 + 1 : 1];
        cast_sample(o, d, radius, t0, t1, cfg.ray_shape, lm, lv);
        if (cfg.disable_integration) { lv[0] = 0.0f; lv[1] = 0.0f; lv[2] = 0.0f; }        /* models.py:228-231 */
      };
      {
      /* P1: four lanes per sample, each 24 of the 96 IPE features (level_fwd_split) */
      const RunConsts rc = run_consts();
      const int i16 = rc.i16, bq = rc.bq;
      const unsigned voff_c = rc.voff_c;
      const int hb = bq >> 1, qq = bq & 1;
      float lm[3], lv[3];
      lift(lm, lv);
      char *xw = Xb + (wave * 16 + i16) * 16;
      const unsigned voff_ipe = blk_voff_add(voff_c, 48 * hb + 24 * qq);
      RN_STAMPW(A, 17);
#pragma clang loop unroll(disable)
      for (int t = 0; t < 12; ++t) {
        unsigned whi, wlo;
        {
          float f[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int kk = 2 * t + u;
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
        /* canonical rows 48 hb + 24 qq + 2 t, + 1: the pair's hi and lo units */
        win_store(aw, voff_ipe, AQ_IPE + 2 * t, 0, whi);
        if (ACT_LO) win_store(aw, voff_ipe, AQ_IPE + 2 * t, 1, wlo);
      }
      }
      RN_STAMPW(A, 18);
      wave_sync();
      RN_STAMPW(A, 3 + phase * 4);
      {
        /* the pipe's lane constants of the spatial section, from here (not from the kernel's entry) */
        const int ln = fresh_lane();
        p.lane = ln;
        p.xps = Xb + ((ln >> 4) * BT + wave * 16 + (ln & 15)) * 16;
      }
      if (phase == 0) {
#pragma unroll
        for (int d = AF; d < SQ_NF; ++d) ar[d] = lds_frag<MM>(p.wbuf + p.cur_off + 1024 + p.lane * 16 + d * 1024);
      }
      const RunConsts rt_ = run_consts();
      const int bq = rt_.bq;
      const unsigned voff_sp = blk_voff_add(rt_.voff_c, 4 * bq), voff_h = blk_voff_add(rt_.voff_c, 2 * bq);
      SqAcc accs[2];
      sq_bias_now(p, accs[1]);
      tq_layer<true, 0, ACT_LO>(p, ar, accs, false, R0, R0, aw, voff_sp, voff_h, AQ_SP, AQ_MASK);
      RN_STAMPW(A, 4 + phase * 4);
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        tq_layer<false, tq_vm_layer<ACT_LO>(), ACT_LO>(p, ar, accs, it == 2, R0, R1, aw, voff_sp, voff_h, AQ_SP + (2 * it + 1) * WIDTH, AQ_MASK + 8 * (2 * it + 1));
        if (it < 3) tq_layer<false, tq_vm_layer<ACT_LO>(), ACT_LO>(p, ar, accs, false, R1, R0, aw, voff_sp, voff_h, AQ_SP + (2 * it + 2) * WIDTH, AQ_MASK + 8 * (2 * it + 2));
      }
      RN_STAMPW(A, 5 + phase * 4);
      {
        /* P3: heads.  Four bottleneck slices (all three products) -> their one-half pair rows of ACT (the directional phase
         * reads them back in its own layout); the scalar block -> LDS HD */
        SqAcc ha[2];
        sq_bias_now(p, ha[1]);
#pragma unroll
        for (int ob = 0; ob < 5; ++ob) {
          SqAcc &acc = ha[(ob + 1) & 1];
          if (ob < 4) {
            if (ob == 0) tq_chunk<SQ_A, true, false, tq_vm_layer<ACT_LO>()>(p, ar, R1, acc, ha[ob & 1]);
            else tq_chunk<SQ_A, true, false, 4>(p, ar, R1, acc, ha[ob & 1]);
            tq_chunk<SQ_B, false, true, 0>(p, ar, R1, acc, ha[ob & 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned pk = e == 0 ? pk_f16(acc.t0[0], acc.t0[1]) : (e == 1 ? pk_f16(acc.t0[2], acc.t0[3])
                                : (e == 2 ? pk_f16(acc.t1[0], acc.t1[1]) : pk_f16(acc.t1[2], acc.t1[3])));
              /* rows 32 ob + 16 (e / 2) + 4 b + 2 (e % 2), + 1 -> unit 16 ob + 8 (e / 2) + 2 b + e % 2 */
              win_store(aw, voff_h, opaque_s(AQ_DIN + 16 * ob), 8 * (e >> 1) + (e & 1), pk);
            }
          } else {
            tq_chunk<SQ_SC, true, false, 4>(p, ar, R1, acc, ha[ob & 1]);
            const RunConsts rh = run_consts();
            int csl = rh.cs + 4 * rh.bq * BT;
            asm volatile("" : "+v"(csl));
            const int bq = rh.bq;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (4 * bq + i < HD_ROWS) HD[i * BT + csl] = acc.t0[i];
          }
        }
        wave_sync();
      }
      RN_STAMPW(A, 6 + phase * 4);
      {
        /* density-gradient normals: seed = W_density through the last ReLU, then layers 7..1 transposed; the IPE rows of layers
         * 5 and 0 through d feature / d mean.  The sign words come back from ACT (this wave's own stores). */
        /* d feature / d mean of this lane's 24 IPE rows into its own columns of the IPE planes (dead since layer 5 of this run;
         * wave-private bytes: the next run's P1 of a wave that runs ahead cannot touch another wave's) */
        const RunConsts rv = run_consts();
        const int i16 = rv.i16, bq = rv.bq, gs = rv.gs;
        const bool vs = rv.vs;
        const unsigned voff_h = blk_voff_add(rv.voff_c, 2 * bq);
        char *dfac = Xb + (wave * 16 + i16) * 16 + bq * 4;
        {
          float lm[3], lv[3];
          lift(lm, lv);
#pragma clang loop unroll(disable)
          for (int v = 0; v < 24; ++v) {
            const int kp = 32 * (v >> 3) + 16 * ((v >> 2) & 1) + (v & 3) + 4 * bq;
            const int hb = kp >= 48 ? 1 : 0;
            const int kk = kp - 48 * hb;                         /* 3 j + axis */
            const int jd = (kk * 43) >> 7;                       /* kk / 3 for kk < 48 */
            const int ax = kk - 3 * jd;
            const float m = ax == 0 ? lm[0] : (ax == 1 ? lm[1] : lm[2]);
            const float vv = ax == 0 ? lv[0] : (ax == 1 ? lv[1] : lv[2]);
            *reinterpret_cast<float *>(dfac + (v >> 1) * (BT * 16) + (v & 1) * ((BT / 2) * 16)) = tq_ipe_dmean(m, vv, jd, hb);
          }
          wave_sync();
        }
        float c, mx;
        {
          const unsigned m7a = win_load(aw, voff_h, opaque_s(AQ_MASK + 56), 0), m7b = win_load(aw, voff_h, opaque_s(AQ_MASK + 56), 1);
          /* this lane's 64 seed values: features 32 s + 16 hf + 4 b + i.  Two walks over the 1 KB vector (L2): the largest
           * entry first, then the scaled split -- instead of 64 staged registers */
          float m = 0.0f;
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const v4f w4 = *reinterpret_cast<const v4f *>(KC + TRC_WD + 32 * s + 16 * hf + 4 * bq);
#pragma unroll
              for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(tq_keep(w4[i], s < 4 ? m7a : m7b, 8 * (s & 3) + 4 * hf + i)));
            }
          m = tq_max4(m);
          c = tq_bound_scale(m, 1.0f);
          mx = m * c;
          const float *KC2 = KC;
          asm volatile("" : "+s"(KC2));
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const v4f w4 = *reinterpret_cast<const v4f *>(KC2 + TRC_WD + 32 * s + 16 * hf + 4 * bq);
#pragma unroll
              for (int q2 = 0; q2 < 2; ++q2) {
                const float x0 = tq_keep(w4[2 * q2], s < 4 ? m7a : m7b, 8 * (s & 3) + 4 * hf + 2 * q2) * c;
                const float x1 = tq_keep(w4[2 * q2 + 1], s < 4 ? m7a : m7b, 8 * (s & 3) + 4 * hf + 2 * q2 + 1) * c;
                unsigned hi, lo;
                split_pair_f16(x0, x1, hi, lo);
                R0[2 * s][2 * hf + q2] = hi;
                R0[2 * s + 1][2 * hf + q2] = lo;
              }
            }
        }
        float ga[3] = {0.0f, 0.0f, 0.0f};
        /* (L7 L6) ([ipe 5] L5 L4) (L3 L2) L1 [ipe 0]: two layers per trip, R0 -> R1 -> R0 */
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {
          const int l = 7 - 2 * it;
          if (it == 1) tq_vjp_ipe(p, ar, R0, dfac, 1.0f / c, ga);
          float rs = tq_bound_scale(tq_max4(mx), KC[TRC_G + TRG_SP + l]);
          c = fminf(fmaxf(c * rs, 0x1p-100f), 0x1p100f);
          tq_vjp_layer(p, ar, R0, R1, aw, voff_h, opaque_s(AQ_MASK + 8 * (l - 1)), rs, mx);
          rs = tq_bound_scale(tq_max4(mx), KC[TRC_G + TRG_SP + l - 1]);
          c = fminf(fmaxf(c * rs, 0x1p-100f), 0x1p100f);
          tq_vjp_layer(p, ar, R1, R0, aw, voff_h, opaque_s(AQ_MASK + 8 * (l - 2)), rs, mx);
        }
        {
          const float rs = tq_bound_scale(tq_max4(mx), KC[TRC_G + TRG_SP + 1]);
          c = fminf(fmaxf(c * rs, 0x1p-100f), 0x1p100f);
          tq_vjp_layer(p, ar, R0, R1, aw, voff_h, opaque_s(AQ_MASK), rs, mx);
        }
        tq_vjp_ipe(p, ar, R1, dfac, 1.0f / c, ga);
        /* ga[a] belongs to axis (a + r) mod 3: back to axes, then the sample's four lanes */
        float gl[3];
        {
          const bool r1 = bq == 1, r2 = bq == 2;
#pragma unroll
          for (int x = 0; x < 3; ++x) gl[x] = tq_sum4(r1 ? ga[(x + 2) % 3] : (r2 ? ga[(x + 1) % 3] : ga[x]));
        }
        const float gx[3] = {-gl[2], -gl[1], -gl[0]};            /* basis^T (octahedron/1: lifted = (-z,-y,-x)) */
        const float ng = sqrtf(fmaxf((gx[0] * gx[0] + gx[1] * gx[1]) + gx[2] * gx[2], EPS32));
        if (bq == 0 && vs) {
#pragma unroll
          for (int i = 0; i < 3; ++i) PS[gs * NP + PS_NORMALS + i] = -(gx[i] / ng);
        }
      }
      RN_STAMPW(A, 7 + phase * 4);
    }
    RN_STAMPW(A, 12);
    /* (round 6) the IDE planes this phase writes (columns 32 w .. 32 w + 31 of k-groups 0..9) are the IPE planes of OTHER waves'
     * runs (columns 16 u .. of the hi / lo halves), where the VJP keeps its d feature / d mean factors and reads the last eight of
     * them BEHIND the final rendezvous of the run: every wave is through with them before any wave overwrites them.  (The window was
     * a few hundred cycles against this phase's ~10 k of arithmetic in front of its first store -- never observed, not excluded.) */
    tq_phase_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) { R0[e] = (v4uu){0, 0, 0, 0}; R1[e] = (v4uu){0, 0, 0, 0}; }
    {
    /* P4: head activations, reflection, IDE (32 samples per wave from here on) */
    const int lane_d = fresh_lane();
    const int h = lane_d >> 5, n = lane_d & 31, col = wave * 32 + n;      /* (shadow the entry values from here to the end of the pass) */
    p.lane = lane_d; p.h = h;
    p.xp = Xb + (h * BT + col) * 16;                                      /* the pipe's lane constants of the directional section */
    int g_w, rl_w; bool valid;
    locate(g_w, rl_w, valid);
    const unsigned voff_d = blk_voff(aw, gs_pass + col, AQ_UNITS, valid);
    const unsigned voff_h2 = blk_voff_add(voff_d, 2 * h), voff_h4 = blk_voff_add(voff_d, 4 * h);
    {
      char *xs = Xb + col * 16;
      SampleHeads sh;
      load_heads(sh);
      float ide[40];
#pragma unroll
      for (int q = 36; q < 40; ++q) ide[q] = 0.0f;
      if (cfg.dir_enc == REFNERF_DIRENC_POSENC) posenc_eval<false, true>(sh.refd[0], sh.refd[1], sh.refd[2], h, [&](int q, float val) { ide[q] = val; });
      else ide_eval<false>(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, h, [&](int q, float val) { ide[q] = val; });
      if (h == 0) ide[36] = sh.dot;
      /* ACT rows 128 + 36 h + q (q < 36) in pairs: unit AQ_DIN + 64 + 18 h + q / 2; half 0 adds (n.v, 0) and the (0, 0) pad pair */
      const unsigned voff_ide = blk_voff_add(voff_d, 18 * h), voff_nv = h == 0 ? voff_d : BLK_NONE;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        v8mm pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = (mm_t)ide[q * 8 + e];
        *reinterpret_cast<v8mm *>(xs + (5 * h + q) * BT * 16) = pk;
        const v4uu pw = __builtin_bit_cast(v4uu, pk);
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          if (8 * q + 2 * e2 < IDE_TERMS) win_store(aw, voff_ide, opaque_s(AQ_DIN + 64 + 4 * q), e2, pw[e2]);
        if (q == 4) { win_store(aw, voff_nv, opaque_s(AQ_DIN + 100), 0, pw[2]); win_store(aw, voff_nv, opaque_s(AQ_DIN + 100), 1, 0u); }
      }
    }
    wave_sync();
    RN_STAMPW(A, 13);
    /* the bottleneck of both runs, back from ACT in the k-step order of the plain BNLDS chunk:
     * dword q of k-step t = rows 32 (t / 2) + 16 (t & 1) + 8 (q / 2) + 4 h + 2 (q & 1), + 1 */
    v4uu bn[8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           /* this wave's own stores have left (same CU: the loads see them) */
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) bn[t][q] = win_load(aw, voff_h2, opaque_s(AQ_DIN + 16 * (t >> 1)), 8 * (t & 1) + 4 * (q >> 1) + (q & 1));
    v8mm (&ad)[AF] = reinterpret_cast<v8mm (&)[AF]>(ar);
    tq_dir_layer<BF_BNLDS, BF_DIR_REAL_KS, 0>(p, ad, 0, R0, bn, R0, aw, voff_h2, voff_h4, AQ_VD, AQ_MASK + 64);
    RN_STAMPW(A, 14);
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      tq_dir_layer<BF_REG, 0, TQ_VM_DIR_LAYER>(p, ad, (it == 2) ? 2 : 0, R0, bn, R1, aw, voff_h2, voff_h4, AQ_VD + (2 * it + 1) * (WIDTH / 2), AQ_MASK + 64 + 8 * (2 * it + 1));
      if (it < 3) tq_dir_layer<BF_REG, 0, TQ_VM_DIR_LAYER>(p, ad, 0, R1, bn, R0, aw, voff_h2, voff_h4, AQ_VD + (2 * it + 2) * (WIDTH / 2), AQ_MASK + 64 + 8 * (2 * it + 2));
    }
    RN_STAMPW(A, 15);
    /* rgb: one slice, [hi][lo] */
    v16f acc;
    tq_bf_chunk<false, BF_REG, 0, true, TQ_VM_DIR_LAYER>(p, ad, R1, bn, acc);
    tq_bf_chunk<false, BF_REG, 0, false, 0>(p, ad, R1, bn, acc);
    float raw_rgb[3];
    /* P6 forms its lane constants again (P4's would ride across the directional trunk in scratch) */
    int lane_w = fresh_lane(), pass_w = pass0;
    asm volatile("" : "+s"(pass_w));
    const int n6 = lane_w & 31, col6 = wave * 32 + n6;
#pragma unroll
    for (int i = 0; i < 3; ++i) raw_rgb[i] = tq_lane_read(acc[i], n6);
    int g6, rl6; bool valid6;
    locate(g6, rl6, valid6);
    if (valid6 && (lane_w >> 5) == 0) {                                               /* P6 */
      SampleHeads sh;
      load_heads(sh);
#pragma unroll
      for (int i = 0; i < 3; ++i) sh.normals[i] = PS[g6 * NP + PS_NORMALS + i];      /* (the VJP's; colour_store writes them back) */
      colour_store<false, NP, 0, true, true>(A, sh, raw_rgb, PS, PX, n_tot, g6, col6);
      /* what the backward needs of the forward: the raw scalar head rows and raw rgb */
      const unsigned voff_d6 = blk_voff(aw, (long long)ray0 * N + pass_w + col6, AQ_UNITS, valid6);
      int ci = col6;
      asm volatile("" : "+v"(ci));
#pragma unroll
      for (int i = 0; i < 11; ++i) win_store(aw, voff_d6, opaque_s(AQ_RAW), i, __builtin_bit_cast(unsigned, HD[i * BT + ci]));
#pragma unroll
      for (int i = 0; i < 3; ++i) win_store(aw, voff_d6, opaque_s(AQ_RAW), 11 + i, __builtin_bit_cast(unsigned, raw_rgb[i]));
    }
    wave_sync();
    history_flush<NP, 0>(A, PS, PX, n_tot, pass_w + wave * 32, wave * 32, (size_t)ray0 * N + pass_w + wave * 32, lane_w);
    RN_STAMPW(A, 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  composite_phase<BF_NW, false, NP>(A, TD, XP, PS, n_tot, ray0, wave, fresh_lane(), reinterpret_cast<float *>(WB), NRM);   /* P7 */
}

__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_train_sq(const LevelArgs A) { level_fwd_train_sq_body<true>(A); }
/* ... for the hi-halves-only weight-gradient GEMM (cfg.wgrad_mode = REFNERF_WGRAD_F16) */
__global__ __launch_bounds__(BF_NTHREADS) void level_fwd_train_sq_h(const LevelArgs A) { level_fwd_train_sq_body<false>(A); }

}  // namespace rn
