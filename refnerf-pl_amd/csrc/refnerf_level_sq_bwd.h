/*
 * refnerf_level_sq_bwd.h -- backward of one level in the parity-grade 16-bit mode on the EVAL kernel's skeleton (round 5;
 * refnerf_sq_layout.h): 8 waves x 32 samples, two waves per SIMD, the transposed weights through the LDS-DMA chunk ring, deltas
 * in registers as packed f16 B fragments of v_mfma_f32_32x32x16_f16.
 *
 * Arithmetic: every transposed layer as TWO products [W^T_hi | W^T_lo] delta (W at 22 bits, the delta of a sample as ONE half
 * after its power-of-two factor -- scripts/exp_train_sq_precision.py: the gradient error of the reference's autograd is
 * unchanged against three products, what matters is the coherent rounding of W).  A layer is then the plain 16-bit chunk
 * pipeline with two chunks per slice and half the registers of a split operand: 32 samples per wave, ONE run per pass.
 * Per-sample factors: before a contraction the sample's deltas are rescaled by a power of two so that
 * max|delta| * G < 2^15, G = the layer's column-sum bound from the image (|W^T delta|_inf <= G |delta|_inf): no pass over
 * the outputs, no overflow for any weights; the outputs then sit 2^3 .. 2^6 below the top of the half's range.
 * Every delta leaves for DELTA as it is produced (the packed halves themselves) with its factor c and the nominal factor
 * kappa per (layer, sample).  Nothing of the forward is recomputed: raw head rows and raw rgb come back from ACT.
 *
 * Restates the autograd of internal/models.py:533-750; oracle: rn_level_backward.  The per-ray part
 * (rendering -> per-sample seeds) stays bwd_seed_kernel of refnerf_level_bwd_f32.h.
 */
#pragma once
#include "refnerf_level_sq_fwd.h"

namespace rn {

struct SqBwdArgs {
  const void *packed;
  refnerf_level_cfg cfg;
  const float *viewdirs;    /* [R,3] */
  long long S;              /* R * N samples */
  int passes;               /* 256-sample passes per workgroup */
  const float *g_s_diffuse, *g_s_specular, *g_s_tint, *g_s_rough;   /* optional per-sample seeds (models.py:731-750), or NULL */
  const float *act;         /* REFNERF_ACT_SQ */
  float *delta;             /* DELTA pair rows + factor units (DQ_*) */
  const float *seeds;       /* [NGS][pitch] per-sample seeds of bwd_seed_kernel: density, rgb[3], n_pred[3] */
  long long pitch;
  long long *prof;
};
constexpr int SQ_NGS = 7;
constexpr int SQB_GI_ROWS = 76;                       /* gradient w.r.t. [IDE 72 | n.v | pad]: rows 128..203 of the dir input */
constexpr int SQB_XE_BYTES = 2 * BT * 16;             /* the 11 scalar head-row deltas as two f16 k-groups */
constexpr int SQB_LDS_BYTES = BF_RING_BYTES + SQB_XE_BYTES + SQB_GI_ROWS * BT * 4;

__device__ __forceinline__ float sq_max2(float m) { return fmaxf(m, __shfl_xor(m, 32, 64)); }

/* acc (fp32 tile of W^T delta, carrying the input's factor) -> recorded ReLU sign bits, the sample's rescale, running max ->
 * two packed f16 B fragments of the next layer.  bit(r) = the bit of register r in the words the caller passes. */
template <typename BitOf>
__device__ __forceinline__ void sq_mask_pack(const v16f &a, v4uu &f0, v4uu &f1, float rs, float &mx, BitOf &&keep) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = keep(a[r], r) * rs;
    mx = fmaxf(mx, fabsf(v[r]));
  }
  /* (pinned per slice: left to itself the optimiser re-associates the running maximum of a layer into ONE tree behind the last
   * slice and keeps all 128 scaled values alive for it -- in scratch, and every reload drains the store queue) */
  asm volatile("" : "+v"(mx));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f0[e] = pk_f16(v[2 * e], v[2 * e + 1]);
    f1[e] = pk_f16(v[8 + 2 * e], v[8 + 2 * e + 1]);
  }
}
/* the eight dwords of k-steps 2 ob, 2 ob + 1 to their pair rows of DELTA (one half per element) */
__device__ __forceinline__ void sq_store_delta(const BlkWin &dw, unsigned voff_h2, int unit, const v4uu &f0, const v4uu &f1) {
#ifdef REFNERF_EXPERIMENT_NO_DELTA   /* timing experiment only: the chains run on real data, their deltas are not written */
  return;
#endif
#ifdef REFNERF_EXPERIMENT_WIDE   /* timing experiment only (wrong layout): the same bytes as two 16-byte stores per lane */
  {
    unsigned vo = (voff_h2 & ~255u) + (voff_h2 & 255u) * 4u;
    asm volatile("" : "+v"(vo));
    __builtin_amdgcn_raw_buffer_store_b128(f0, dw.rs, vo, unit * 256, REFNERF_SQ_STREAM_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(f1, dw.rs, vo + 2048u, unit * 256, REFNERF_SQ_STREAM_AUX);
    return;
  }
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    win_store(dw, voff_h2, unit, 4 * (q >> 1) + (q & 1), f0[q]);
    win_store(dw, voff_h2, unit, 8 + 4 * (q >> 1) + (q & 1), f1[q]);
  }
}
/* factor units of layer id `lid`: c (0 for a sample without any delta there: it takes no part in the layer's smallest
 * kappa) and kappa = c 2^(15 - e(m)), m = the sample's largest stored |delta| (both half-waves hold the same values) */
__device__ __forceinline__ void sq_store_factors(const BlkWin &dw, unsigned voff_f, int lid, float c, float m) {
  const bool live = m > 0.0f;
  const float kappa = c * tq_bound_scale(m, 1.0f);
  win_store(dw, voff_f, opaque_s(DQ_C + lid), 0, __builtin_bit_cast(unsigned, live ? c : 0.0f));
  win_store(dw, voff_f, opaque_s(DQ_K + lid), 0, __builtin_bit_cast(unsigned, live ? kappa : 0.0f));
}
__device__ __forceinline__ float sq_clampc(float c) { return fminf(fmaxf(c, 0x1p-100f), 0x1p100f); }

/* One transposed 256 -> 256 layer on 32 samples: out = mask (W^T in) rs, 8 slices x [hi chunk][lo chunk].
 * VMK0: vector-memory operations in front of the first chunk; every slice leaves 8 DELTA dwords behind its lo chunk. */
template <int VMK0, typename BitOf>
__device__ __forceinline__ void sq_bwd_layer(Pipe &p, MmF16::v8 (&a)[AF], const v4uu (&in)[16], v4uu (&out)[16], const BlkWin &dw, unsigned voff_h2,
                                             int dunit, float rs, float &mx, BitOf &&keep) {
  const v4uu (&nobn)[8] = reinterpret_cast<const v4uu (&)[8]>(in);
  mx = 0.0f;
  dunit = opaque_s(dunit);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (ob == 0) tq_bf_chunk<true, BF_REG, 0, false, VMK0>(p, a, in, nobn, acc);
    else tq_bf_chunk<true, BF_REG, 0, false, 8>(p, a, in, nobn, acc);
    tq_bf_chunk<true, BF_REG, 0, false, 0>(p, a, in, nobn, acc);
    sq_mask_pack(acc, out[2 * ob], out[2 * ob + 1], rs, mx, [&](float x, int r) { return keep(x, ob, r); });
    sq_store_delta(dw, voff_h2, dunit + 16 * ob, out[2 * ob], out[2 * ob + 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void level_bwd_sq_body(const SqBwdArgs &A) {
  typedef MmF16 MM;
  typedef MM::v8 v8mm;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const refnerf_level_cfg &cfg = A.cfg;
  const int N = cfg.n_samples;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, n = lane & 31;
  const int col = wave * 32 + n;

  char *WB = reinterpret_cast<char *>(smem);                 /* 3 x 17 KB chunk ring                          */
  char *XE = WB + BF_RING_BYTES;                             /* [2 k-groups][BT][16 B]: scalar head-row deltas */
  float *GI = reinterpret_cast<float *>(XE + SQB_XE_BYTES);  /* [76][BT]: gradient w.r.t. the dir encodings    */
  const float *KC = reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.packed) + TR_CONST_OFF);

  const long long gs_wg = (long long)blockIdx.x * A.passes * BT;
  int n_pass = A.passes;
  if (gs_wg + (long long)n_pass * BT > A.S) n_pass = (int)((A.S - gs_wg + BT - 1) / BT);

  Pipe p;
  p.src = reinterpret_cast<const char *>(A.packed) + (size_t)TR_BWD0 * BF_CHUNK_BYTES + wave * 3072 + lane * 16;
  p.src_end = nullptr;
  p.wbuf = WB;
  p.xp = XE + (h * BT + col) * 16;
  p.xps = nullptr;
  p.seq = 0;
  p.cur_off = 0; p.nxt_off = BF_CHUNK_BYTES; p.fil_off = 2 * BF_CHUNK_BYTES;
  p.dma_left = n_pass * TR_BWD;
  p.lane = lane; p.wave = wave; p.h = h;
  p.t_vm = 0; p.t_bar = 0;
  RN_STAMPW(A, 0);
  tq_issue<true>(p, p.cur_off);
  tq_issue<true>(p, p.nxt_off);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifndef REFNERF_BF_NOPRIO
  if (wave >= BF_NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  v4uu R0[16], R1[16];
  v8mm ad[AF];
#pragma unroll
  for (int d = 0; d < AF; ++d) ad[d] = lds_frag<MM>(WB + 1024 + lane * 16 + d * 1024);

#pragma unroll 1
  for (int pass = 0; pass < n_pass; ++pass) {
    const long long gs_pass = gs_wg + (long long)pass * BT;
    {
      /* a wave whose 32 samples are all past the end keeps the stream going */
      if (gs_pass + wave * 32 >= A.S) { tq_idle_pass<true>(p); continue; }
    }
    /* (lane constants of the pass from a lane index formed here: carried from the kernel's entry they sit in scratch) */
    const int lane_p = fresh_lane();
    const int lane = lane_p;                /* (shadows the entry value for the cycle stamps of the pass) */
    int col_v = wave * 32 + (lane_p & 31), h_v = lane_p >> 5;
    asm volatile("" : "+v"(col_v), "+v"(h_v));
    const long long gs = gs_pass + col_v;
    const bool valid = gs < A.S;
    const void *act_l = A.act;
    void *del_l = A.delta;
    asm volatile("" : "+s"(act_l), "+s"(del_l));
    const BlkWin aw = blk_window(act_l, gs_pass, AQ_UNITS), dw = blk_window(del_l, gs_pass, DQ_UNITS);
    const unsigned voff_a = blk_voff(aw, gs, AQ_UNITS, valid), voff_d = blk_voff(dw, gs, DQ_UNITS, valid);
    const unsigned voff_dh2 = blk_voff_add(voff_d, 2 * h_v);                 /* pair rows of this half-wave: + 2 h units   */
    const unsigned voff_f = h_v == 0 ? voff_d : BLK_NONE;                    /* per-sample units: half 0 stores            */
    RN_STAMPW(A, 1);

    /* ===== per-sample head: what the forward left in ACT, the colour head backward (models.py:699-729) =====
     * Evaluated TWICE -- here for the seed of the directional chain (g_raw_rgb), and again in front of the IDE backward -- from
     * 24 reloaded values (L2) and ~300 VALU, instead of ~45 registers carried across both trunks (at 256 registers per lane that
     * was 330 spilled VGPRs and 0.9 KB / lane of scratch). */
    struct HeadState { SampleHeads sh; float g_raw_rgb[3], g_tint[3], g_raw_diff[3], gsv[SQ_NGS], v[3], raw_density, raw_rough; };
    auto head_state = [&](HeadState &H) {
      int cv2 = col_v;
      asm volatile("" : "+v"(cv2));
      const long long gs2 = gs_pass + cv2;
      const bool valid2 = gs2 < A.S;
      const unsigned va = blk_voff(aw, gs2, AQ_UNITS, valid2);
      float gp[3], raw_dif[3], raw_tint[3], raw_rgb[3];
      const int as = opaque_s(AQ_RAW);
      H.raw_density = __builtin_bit_cast(float, win_load(aw, va, as, 0));
      H.raw_rough = __builtin_bit_cast(float, win_load(aw, va, as, 4));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        gp[i] = __builtin_bit_cast(float, win_load(aw, va, as, 1 + i));
        raw_dif[i] = __builtin_bit_cast(float, win_load(aw, va, as, 5 + i));
        raw_tint[i] = __builtin_bit_cast(float, win_load(aw, va, as, 8 + i));
        raw_rgb[i] = __builtin_bit_cast(float, win_load(aw, va, as, 11 + i));
      }
      const long long gsc = valid2 ? gs2 : 0;
      const long long ray = gsc / N;
#pragma unroll
      for (int i = 0; i < 3; ++i) H.v[i] = A.viewdirs[ray * 3 + i];
#pragma unroll
      for (int i = 0; i < SQ_NGS; ++i) H.gsv[i] = valid2 ? A.seeds[(size_t)i * A.pitch + gsc] : 0.0f;
      sample_heads(cfg, H.raw_density, gp, H.raw_rough, raw_dif, raw_tint, H.v, H.sh);
      float sg[3], dl[3], colr[3], g_col[3];
      const float pad_scale = (float)(1.0 + 2.0 * (double)cfg.rgb_padding);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        sg[i] = sigmoid_t(cfg.rgb_premultiplier * raw_rgb[i] + cfg.rgb_bias);
        dl[i] = sigmoid_t(H.sh.raw_dif[i] - LOG3_F);
        colr[i] = H.sh.tint[i] * sg[i] + dl[i];
        g_col[i] = H.gsv[1 + i] * pad_scale;
      }
      if (cfg.srgb_mapping) colour_map_backward(colr, cfg.srgb_mapping_normalization != 0, true, g_col);
      float g_dl[3], g_sp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { g_dl[i] = g_col[i]; g_sp[i] = g_col[i]; }
      if (A.g_s_diffuse || A.g_s_specular) {
        float e_dl[3], e_sp[3], spl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          /* (gsc is clamped: the load itself is unconditional per lane, a wave-uniform branch on the pointer only) */
          const float ld = A.g_s_diffuse ? A.g_s_diffuse[gsc * 3 + i] : 0.0f, ls = A.g_s_specular ? A.g_s_specular[gsc * 3 + i] : 0.0f;
          e_dl[i] = valid2 ? ld : 0.0f;
          e_sp[i] = valid2 ? ls : 0.0f;
          spl[i] = H.sh.tint[i] * sg[i];
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
        const float lt = A.g_s_tint ? A.g_s_tint[gsc * 3 + i] : 0.0f;
        H.g_tint[i] = g_sp[i] * sg[i] + (valid2 ? lt : 0.0f);
        H.g_raw_rgb[i] = (g_sp[i] * H.sh.tint[i]) * sg[i] * (1.0f - sg[i]) * cfg.rgb_premultiplier;
        H.g_raw_diff[i] = g_dl[i] * dl[i] * (1.0f - dl[i]);
      }
    };
    float g_raw_rgb[3];
    {
      HeadState H;
      head_state(H);
#pragma unroll
      for (int i = 0; i < 3; ++i) g_raw_rgb[i] = H.g_raw_rgb[i];
    }
    RN_STAMPW(A, 2);
    float c, mx;
    {
      /* rgb rows of DELTA, then the seed of the directional chain: W_rgb^T g_raw_rgb through the last ReLU */
      const float mg = fmaxf(fmaxf(fabsf(g_raw_rgb[0]), fabsf(g_raw_rgb[1])), fabsf(g_raw_rgb[2]));
      const float cr = tq_bound_scale(mg, 1.0f);
      win_store(dw, voff_f, opaque_s(DEL_RGB / 2), 0, pk_f16(g_raw_rgb[0] * cr, g_raw_rgb[1] * cr));
      win_store(dw, voff_f, opaque_s(DEL_RGB / 2), 1, pk_f16(g_raw_rgb[2] * cr, 0.0f));
      sq_store_factors(dw, voff_f, 17, cr, mg * cr);
      unsigned M7[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) M7[q] = win_load(aw, blk_voff_add(voff_a, 4 * h_v), opaque_s(AQ_MASK + 64 + 56), q);
      c = tq_bound_scale(mg, KC[TRC_G + TRG_RGB]);
      mx = 0.0f;
      const float gr0 = g_raw_rgb[0] * c, gr1 = g_raw_rgb[1] * c, gr2 = g_raw_rgb[2] * c;
      const float *W = KC + TRC_WRGB + 4 * h_v;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
        float vv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f w0 = *reinterpret_cast<const v4f *>(W + 32 * ob + 8 * q), w1 = *reinterpret_cast<const v4f *>(W + WIDTH + 32 * ob + 8 * q),
                    w2 = *reinterpret_cast<const v4f *>(W + 2 * WIDTH + 32 * ob + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i;
            const float x = tq_keep((w0[i] * gr0 + w1[i] * gr1) + w2[i] * gr2, M7[ob >> 1], 16 * (ob & 1) + r);
            vv[r] = x;
            mx = fmaxf(mx, fabsf(x));
          }
        }
        asm volatile("" : "+v"(mx));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          R0[2 * ob][e] = pk_f16(vv[2 * e], vv[2 * e + 1]);
          R0[2 * ob + 1][e] = pk_f16(vv[8 + 2 * e], vv[8 + 2 * e + 1]);
        }
        sq_store_delta(dw, voff_dh2, opaque_s((DEL_VD + 7 * WIDTH) / 2 + 16 * ob), R0[2 * ob], R0[2 * ob + 1]);
      }
      mx = sq_max2(mx);
      sq_store_factors(dw, voff_f, 9 + 7, c, mx);
    }
    RN_STAMPW(A, 3);
    /* ===== directional layers 7..1 (R0 -> R1 -> R0 ...); layer 5's delta is read back for the dir-input rows below ===== */
    float c5 = 1.0f, m5 = 0.0f;
    {
      const unsigned voff_m4 = blk_voff_add(voff_a, 4 * h_v);
      auto dir_layer = [&](auto VM, const v4uu (&in)[16], v4uu (&out)[16], int l) {
        /* sign words of layer l - 1 (four per half-wave), the rescale from the input's largest entry and the layer's bound */
        unsigned M[4];
        const int mu = opaque_s(AQ_MASK + 64 + 8 * (l - 1));
#pragma unroll
        for (int q = 0; q < 4; ++q) M[q] = win_load(aw, voff_m4, mu, q);
        const float rs = tq_bound_scale(mx, KC[TRC_G + TRG_VD + l]);
        c = sq_clampc(c * rs);
        sq_bwd_layer<decltype(VM)::value>(p, ad, in, out, dw, voff_dh2, (DEL_VD + (l - 1) * WIDTH) / 2, rs, mx,
                                          [&](float x, int ob, int r) { return tq_keep(x, M[ob >> 1], 16 * (ob & 1) + r); });
        mx = sq_max2(mx);
        sq_store_factors(dw, voff_f, 9 + l - 1, c, mx);
      };
      typedef std::integral_constant<int, 8 + 2 + 4> VML;     /* 8 delta dwords + 2 factor units + 4 sign words in front of a layer */
      typedef std::integral_constant<int, 0> VM0;
      dir_layer(VM0(), R0, R1, 7);
#pragma unroll 1
      for (int it = 0; it < 3; ++it) {
        dir_layer(VML(), R1, R0, 6 - 2 * it);
        if (it == 0) { c5 = c; m5 = mx; }                     /* R0 now holds delta_5 (its pair rows are in DELTA) */
        dir_layer(VML(), R0, R1, 5 - 2 * it);
      }
      /* seven layers: delta_0 sits in R1 */
    }
    RN_STAMPW(A, 4);
    float c_bn, mxb = 0.0f;
    v4uu hbn[8];
    {
      /* ===== the 204 dir-input rows: W5[:, 256:]^T delta_5 + W0^T delta_0, 7 slices x [layer 5 hi lo | layer 0 hi lo] =====
       * delta_5 comes back from DELTA (this lane's own dwords) into R0; the two terms carry different factors, so each slice
       * keeps two accumulators and joins them in fp32.  Rows 0..127 = the bottleneck's deltas (head rows 0..127): packed with
       * a factor from the bound |.| <= G5 m5 / c5 + G0 m0 / c0 into the B fragments of heads^T; rows 128..203 -> LDS GI. */
      const float c0 = c, m0 = mx;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      {
        const int du = opaque_s((DEL_VD + 5 * WIDTH) / 2);
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) R0[t][q] = win_load(dw, voff_dh2, du + 16 * (t >> 1), 8 * (t & 1) + 4 * (q >> 1) + (q & 1));
      }
      const float bnd = KC[TRC_G + TRG_VD5_DIN] * (m5 / c5) + KC[TRC_G + TRG_VD0] * (m0 / c0);
      /* (a sample without any delta in the directional trunk -- zero weight, dead trunk -- puts no constraint on the head rows'
       * common factor: the scalar rows then sit at the top of the half's range as in every other sample) */
      c_bn = bnd > 0.0f ? tq_bound_scale(bnd, 1.0f) : 0x1p100f;
      const float ia = c_bn / c5, ib = c_bn / c0, ja = 1.0f / c5, jb = 1.0f / c0;
      const v4uu (&nobn)[8] = reinterpret_cast<const v4uu (&)[8]>(R0);
#pragma unroll
      for (int ob = 0; ob < DIN_BLOCKS; ++ob) {
        v16f accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.0f; accB[r] = 0.0f; }
        if (ob == 0) tq_bf_chunk<true, BF_REG, 0, false, 0>(p, ad, R0, nobn, accA);
        else tq_bf_chunk<true, BF_REG, 0, false, 0>(p, ad, R0, nobn, accA);
        tq_bf_chunk<true, BF_REG, 0, false, 0>(p, ad, R0, nobn, accA);
        tq_bf_chunk<true, BF_REG, 0, false, 0>(p, ad, R1, nobn, accB);
        tq_bf_chunk<true, BF_REG, 0, false, 0>(p, ad, R1, nobn, accB);
        if (ob < 4) {
          float vv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) { vv[r] = accA[r] * ia + accB[r] * ib; mxb = fmaxf(mxb, fabsf(vv[r])); }
          asm volatile("" : "+v"(mxb));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            /* (pinned: MachineSink otherwise moves the conversions to their first use behind the IDE backward and the sixteen
             * fp32 values of every slice wait for them in scratch -- 64 of the 160 spilled registers of round 5) */
            unsigned q0 = pk_f16(vv[2 * e], vv[2 * e + 1]), q1 = pk_f16(vv[8 + 2 * e], vv[8 + 2 * e + 1]);
            asm volatile("" : "+v"(q0), "+v"(q1));
            hbn[2 * ob][e] = q0;
            hbn[2 * ob + 1][e] = q1;
          }
        } else {
          int cg = col_v + 4 * h_v * BT;
          asm volatile("" : "+v"(cg));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * (ob - 4) + (r & 3) + 8 * (r >> 2);          /* + 4 h: rides in cg */
            if (row + 4 * h_v < SQB_GI_ROWS) GI[row * BT + cg] = accA[r] * ja + accB[r] * jb;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      mxb = sq_max2(mxb);
    }
    wave_sync();
    RN_STAMPW(A, 5);
    float c_h;
    {
      /* ===== IDE, reflection, predicted normal, head activations backward (models.py:611-686) ===== */
      HeadState H;
      head_state(H);
      __builtin_amdgcn_sched_barrier(0);      /* (the phase is 11 k VALU instructions: one scheduling region of it fills 256 registers by itself) */
      const SampleHeads &sh = H.sh;
      const float (&v)[3] = H.v;
      const float (&gsv)[SQ_NGS] = H.gsv;
      const float (&g_tint)[3] = H.g_tint;
      const float (&g_raw_diff)[3] = H.g_raw_diff;
      const float raw_density = H.raw_density, raw_rough = H.raw_rough;
      float g_ref[3], g_rough;
      int ci = col_v;
      asm volatile("" : "+v"(ci));
      /* (the read's address is tied to the recurrence value it is needed for: the 72 reads stay where they are used instead of
       *  being hoisted in front of the recurrences and parked in scratch) */
      auto gq = [&](int q, float &dep) {
        int c2 = ci;
        asm volatile("" : "+v"(c2), "+v"(dep));
        return GI[q * BT + c2];
      };
      if (cfg.dir_enc == REFNERF_DIRENC_POSENC) { posenc_grad(sh.refd[0], sh.refd[1], sh.refd[2], gq, g_ref); g_rough = 0.0f; }
      else ide_grad(sh.refd[0], sh.refd[1], sh.refd[2], sh.rough, gq, g_ref, g_rough);
      __builtin_amdgcn_sched_barrier(0);
      const float g_dot = GI[IDE_DIM * BT + ci];
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
      const float livef = (nrm2 > EPS32) ? 1.0f : (nrm2 == EPS32 ? 0.5f : 0.0f);
#pragma unroll
      for (int i = 0; i < 3; ++i) g_gp[i] = -(g_np[i] / rsq - livef * sh.gp[i] * gdotg / (s * rsq));
      if (valid && A.g_s_rough) g_rough += A.g_s_rough[gs];
      const float g_raw_rough = g_rough * softplus_grad(raw_rough + cfg.roughness_bias);
      const float g_raw_density = gsv[0] * softplus_grad(raw_density + cfg.density_bias);
      float hrow[16];
      hrow[0] = g_raw_density;
      hrow[4] = g_raw_rough;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        hrow[1 + i] = g_gp[i];
        hrow[5 + i] = g_raw_diff[i];
        hrow[8 + i] = g_tint[i] * sh.tint[i] * (1.0f - sh.tint[i]);
      }
#pragma unroll
      for (int i = 11; i < 16; ++i) hrow[i] = 0.0f;
      float mh = 0.0f;
#pragma unroll
      for (int i = 0; i < 11; ++i) mh = fmaxf(mh, fabsf(hrow[i]));
      /* ONE factor for the 139 head rows: the smaller of the bottleneck's and the scalars' (neither may overflow) */
      const float c_sc = tq_bound_scale(mh, 1.0f);
      c_h = fminf(c_bn, c_sc);
      const float dn = c_h / c_bn;                            /* a power of two <= 1 */
      if (dn != 1.0f) {
        const unsigned dn2 = pk_f16(dn, dn);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned w = hbn[t][q];         /* (a scalar copy first: __builtin_bit_cast on an ext-vector ELEMENT reads element 0) */
            const MM::v2 z = __builtin_bit_cast(MM::v2, w) * __builtin_bit_cast(MM::v2, dn2);
            hbn[t][q] = __builtin_bit_cast(unsigned, z);
          }
      }
      const float mall = fmaxf(mxb * dn, mh * c_h);
      /* head rows of DELTA: bottleneck pair rows 0..63 (8 k-steps x 4 dwords per half-wave), scalar pair rows 64..71 */
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          win_store(dw, voff_dh2, opaque_s(DEL_HEADS / 2 + 16 * (t >> 1)), 8 * (t & 1) + 4 * (q >> 1) + (q & 1), hbn[t][q]);
      {
        /* this half-wave's eight scalar rows 128 + 8 h ..: four dwords (pair rows 64 + 4 h ..) and its f16 k-group for heads^T */
        unsigned pw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = h_v ? hrow[8 + 2 * e] : hrow[2 * e], x1 = h_v ? hrow[8 + 2 * e + 1] : hrow[2 * e + 1];
          pw[e] = pk_f16(x0 * c_h, x1 * c_h);
        }
        const unsigned voff_s = blk_voff_add(voff_d, 4 * h_v);
#pragma unroll
        for (int e = 0; e < 4; ++e) win_store(dw, voff_s, opaque_s(DEL_HEADS / 2 + 64), e, pw[e]);
        *reinterpret_cast<v4uu *>(XE + (h_v * BT + ci) * 16) = (v4uu){pw[0], pw[1], pw[2], pw[3]};
      }
      sq_store_factors(dw, voff_f, 8, c_h, mall);
      mx = mall;
      c = c_h;
    }
    wave_sync();
    RN_STAMPW(A, 6);
    {
      /* ===== heads^T (139 -> 256: K = the bottleneck fragments + the scalar k-group from LDS), then spatial layers 7..1 =====
       * sign words of a spatial layer: the forward's 16-sample lanes (b, n) hold features 4 b .. 4 b + 3 (+ 16) of every 32:
       * this half-wave's rows come from b = h (registers 0-3, 8-11) and b = h + 2 (4-7, 12-15): two units, two words each */
      const unsigned voff_ma = blk_voff_add(voff_a, 2 * h_v), voff_mb = blk_voff_add(voff_a, 2 * h_v + 4);
      unsigned Ma[2], Mb[2];
      auto load_sp_mask = [&](int l) {
        const int mu = opaque_s(AQ_MASK + 8 * l);
        Ma[0] = win_load(aw, voff_ma, mu, 0); Ma[1] = win_load(aw, voff_ma, mu, 1);
        Mb[0] = win_load(aw, voff_mb, mu, 0); Mb[1] = win_load(aw, voff_mb, mu, 1);
      };
      auto keep_sp = [&](float x, int ob, int r) {
        /* feature 32 ob + (r & 3) + 8 (r >> 2) + 4 h: b = h + 2 ((r >> 2) & 1), i = 4 (r >> 3) + (r & 3); bit 8 (ob % 4) + i of word ob / 4 */
        const unsigned w = ((r >> 2) & 1) ? Mb[ob >> 2] : Ma[ob >> 2];
        return tq_keep(x, w, 8 * (ob & 3) + 4 * (r >> 3) + (r & 3));
      };
      load_sp_mask(7);
      {
        const float rs = tq_bound_scale(mx, KC[TRC_G + TRG_HEADS]);
        c = sq_clampc(c * rs);
        mx = 0.0f;
        const int du = opaque_s((DEL_SP + 7 * WIDTH) / 2);
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
          v16f acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
          if (ob == 0) tq_bf_chunk<true, BF_BNLDS, 1, false, 0>(p, ad, R0, hbn, acc);
          else tq_bf_chunk<true, BF_BNLDS, 1, false, 8>(p, ad, R0, hbn, acc);
          tq_bf_chunk<true, BF_BNLDS, 1, false, 0>(p, ad, R0, hbn, acc);
          sq_mask_pack(acc, R1[2 * ob], R1[2 * ob + 1], rs, mx, [&](float x, int r) { return keep_sp(x, ob, r); });
          sq_store_delta(dw, voff_dh2, du + 16 * ob, R1[2 * ob], R1[2 * ob + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        mx = sq_max2(mx);
        sq_store_factors(dw, voff_f, 7, c, mx);
      }
      RN_STAMPW(A, 7);
      auto sp_layer = [&](const v4uu (&in)[16], v4uu (&out)[16], int l) {
        load_sp_mask(l - 1);
        const float rs = tq_bound_scale(mx, KC[TRC_G + TRG_SP + l]);
        c = sq_clampc(c * rs);
        sq_bwd_layer<8 + 2 + 4>(p, ad, in, out, dw, voff_dh2, (DEL_SP + (l - 1) * WIDTH) / 2, rs, mx, keep_sp);
        mx = sq_max2(mx);
        sq_store_factors(dw, voff_f, l - 1, c, mx);
      };
      sp_layer(R1, R0, 7);
#pragma unroll 1
      for (int it = 0; it < 3; ++it) {
        sp_layer(R0, R1, 6 - 2 * it);
        sp_layer(R1, R0, 5 - 2 * it);
      }
    }
    RN_STAMPW(A, 8);
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int lane = fresh_lane();          /* (shadows the entry value: that one then dies in front of the pass loop) */
    RN_STAMPW(A, 9);
#ifdef REFNERF_PROF_WAITS
    if (A.prof && blockIdx.x == (gridDim.x >> 1) && lane == 0) { A.prof[wave * 32 + 20] = p.t_vm; A.prof[wave * 32 + 21] = p.t_bar; }
#endif
  }
}

__global__ __launch_bounds__(BF_NTHREADS) void level_bwd_sq(const SqBwdArgs A) { level_bwd_sq_body(A); }

}  // namespace rn
