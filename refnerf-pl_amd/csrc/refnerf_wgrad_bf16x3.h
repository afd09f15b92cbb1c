/*
 * refnerf_wgrad_bf16x3.h -- the weight-gradient contraction of refnerf_wgrad.h
 * on the bf16 matrix cores at (near) fp32 accuracy.
 *
 * dW[o][k] = sum_s DELTA[o][s] * ACT[k][s] over 5e5 samples is, on
 * v_mfma_f32_32x32x2_f32, MFMA-bound (12.3 ms per level at C2 for 18 GB of
 * operands).  Here every fp32 operand is split on the fly into two bf16 values,
 * x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits), and the
 * product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation: 3 MFMAs at 16x the fp32-MFMA rate, relative error per
 * product 2^-16 (the dropped lo*lo term) -- below the summation-order noise of
 * an fp32 sum over 1e5 samples.  The contraction becomes HBM-bound, so the
 * second half of the design is traffic: the workgroup -> (tile, slice) mapping
 * puts all tiles of one sample slice on ONE XCD (workgroup ids are dealt
 * round-robin to the 8 XCDs), so the DELTA / ACT row blocks shared by the tiles
 * of a layer are fetched from HBM once and re-read from that XCD's L2.
 *
 * Same job table, split-K slices, PART layout and fixed-order reduction as
 * refnerf_wgrad.h: bit-reproducible, no atomics.
 */
#pragma once
#include "refnerf_level_bf16.h"
#include "refnerf_wgrad.h"

namespace rn {

constexpr int WB_KT = 64;                      /* samples per k-step */
constexpr int WB_ROW = WB_KT * 2 + 16;         /* LDS row pitch in bytes: 144 -> conflict-free ds_read_b128 */
constexpr int WB_TILE = WG_TM * WB_ROW;        /* one 128-row bf16 operand tile: 18 KB */
constexpr int WB_LDS = 4 * WB_TILE;            /* D_hi, D_lo, A_hi, A_lo */
/* a bf16 operand has no low tile: 72 / 54 / 36 KB per workgroup */
constexpr int wb_lds(bool d16, bool a16) { return (2 + (d16 ? 0 : 1) + (a16 ? 0 : 1)) * WB_TILE; }

/* x0, x1 -> packed bf16 pair of the leading 8 mantissa bits and of the next 8 */
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = cvt_pk_bf16(x0 - h0, x1 - h1);
}

/* grid = 8 * ceil(slices / 8) * WJOBS.tiles workgroups of 256 threads (waves 2x2 over the 128x128 tile) */
/* D16 / A16: that operand is a matrix of bf16 rows (written by the bf16-chain kernels): read as is, its low half is 0 */
/* EXT: the tail job table of a general IPE basis (WJOBS_EXT; partials [slices][EXT_PARAMS]) */
template <bool D16, bool A16, bool EXT = false>
__global__ __launch_bounds__(256) void wgrad_bf16x3_kernel(const WgradArgs A, int slices) {
  constexpr const WJobs &JT = EXT ? WJOBS_EXT : WJOBS;
  extern __shared__ __attribute__((aligned(16))) char wbs[];
  char *Dh = wbs, *Dl = wbs + WB_TILE, *Ah = wbs + (D16 ? 1 : 2) * WB_TILE, *Al = Ah + WB_TILE;   /* Dl / Al: only for fp32 operands */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  /* XCD-aware decode: id % 8 = XCD; that XCD walks the tiles of slices xcd, xcd + 8, ... one slice at a time */
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile = q % JT.tiles, slice = (q / JT.tiles) * 8 + xcd;
  if (slice >= slices) return;
  int ji = 0;
#pragma unroll 1
  for (int j = 1; j < JT.n; ++j) if (tile >= JT.job[j].tile0) ji = j;
  const WJob J = JT.job[ji];
  const int tl = tile - J.tile0;
  const int tm = tl / J.tiles_n, tn = tl - tm * J.tiles_n;
  const long long k_begin = (long long)slice * A.k_per_slice;
  long long k_end = k_begin + A.k_per_slice;
  const long long s_pad = (A.S + WB_KT - 1) / WB_KT * WB_KT;   /* <= pitch; columns >= S hold zeros */
  if (k_end > s_pad) k_end = s_pad;

  v16f acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;    /* loader: 8 rows of the tile per thread, 4 samples at lc4 */
  /* fp32 operand: rows lrow + 16p.  bf16 operand (rows stored in pairs, refnerf_level_f32.h elem_index): the thread owns
   * four PAIRS of rows -- one 16-B load brings 4 samples of both rows; p = 2*pp + half.  The pair index is a bit
   * permutation of lrow that puts the two row groups of a half-wave 8 rows (32 LDS banks) apart. */
  const int lpair = (lrow & 8) | ((lrow & 1) << 2) | ((lrow >> 1) & 3);
  auto tile_row = [&](int p, bool h16) { return h16 ? 2 * lpair + 32 * (p >> 1) + (p & 1) : lrow + 16 * p; };
  const char *dp[8], *ap[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int orow = tm * WG_TM + tile_row(p, D16), irow = tn * WG_TN + tile_row(p, A16);
    /* (all row origins of the job table are even, so a tile's row pairs are the matrices' row pairs) */
    /* blocked rows: unit u of the 64-sample block that starts at sample k0 is at dword k0 * units + u * 64 */
    if constexpr (D16) dp[p] = (orow < J.n_out) ? reinterpret_cast<const char *>(A.delta) + ((long long)((J.d_row + orow) >> 1) * RB + lc4) * 4 : nullptr;
    else dp[p] = (orow < J.n_out) ? reinterpret_cast<const char *>(A.delta) + ((long long)(J.d_row + orow) * RB + lc4) * 4 : nullptr;
    if constexpr (A16) ap[p] = (irow < J.n_in) ? reinterpret_cast<const char *>(A.act) + ((long long)((J.a_row + irow) >> 1) * RB + lc4) * 4 : nullptr;
    else ap[p] = (irow < J.n_in) ? reinterpret_cast<const char *>(A.act) + ((long long)(J.a_row + irow) * RB + lc4) * 4 : nullptr;
  }
  static_assert(WB_KT == RB, "one k-step = one 64-sample block of the operand matrices");
  const long long dstep = (long long)A.d_units * 4, astep = (long long)A.a_units * 4;   /* bytes per sample of k0 */
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  /* 4 samples as raw dwords: of one fp32 row, or of a PAIR of bf16 rows (dword = {row 2j | row 2j+1 << 16}; held in the
   * even slot, the odd slot stays unused).  Kept as integers: a packed bf16 pair is not a well-formed float (it may
   * look like a denormal) and must not pass through float registers' canonicalisation */
  v4u dv[8], av[8];
  auto fetch = [&](long long k0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (!(D16 && (p & 1))) {
        v4u x = {0u, 0u, 0u, 0u};
        if (dp[p]) x = *reinterpret_cast<const v4u *>(dp[p] + k0 * dstep);
        dv[p] = x;
      }
      if (!(A16 && (p & 1))) {
        v4u y = {0u, 0u, 0u, 0u};
        if (ap[p]) y = *reinterpret_cast<const v4u *>(ap[p] + k0 * astep);
        av[p] = y;
      }
    }
  };
  /* samples (0,1) and (2,3) of the even (odd) row of a pair as packed bf16 pairs */
  auto unpair = [](const v4u q, int half, bool live, unsigned &s01, unsigned &s23) {
    const unsigned sel = half ? 0x07060302u : 0x05040100u;
    s01 = live ? __builtin_amdgcn_perm(q[1], q[0], sel) : 0u;
    s23 = live ? __builtin_amdgcn_perm(q[3], q[2], sel) : 0u;
  };
  if (k_begin < k_end) fetch(k_begin);
  for (long long k0 = k_begin; k0 < k_end; k0 += WB_KT) {
    __syncthreads();                                   /* previous tile fully consumed */
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int offd = tile_row(p, D16) * WB_ROW + lc4 * 2, offa = tile_row(p, A16) * WB_ROW + lc4 * 2;
      unsigned h0, l0, h1, l1;
      auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
      if constexpr (D16) {
        unpair(dv[p & ~1], p & 1, dp[p] != nullptr, h0, h1); l0 = 0u; l1 = 0u;
        bsum[p] += (__builtin_bit_cast(float, h0 << 16) + __builtin_bit_cast(float, h0 & 0xffff0000u)) +
                   (__builtin_bit_cast(float, h1 << 16) + __builtin_bit_cast(float, h1 & 0xffff0000u));
      } else {
        split_pair(f(dv[p][0]), f(dv[p][1]), h0, l0);
        split_pair(f(dv[p][2]), f(dv[p][3]), h1, l1);
        bsum[p] += (f(dv[p][0]) + f(dv[p][1])) + (f(dv[p][2]) + f(dv[p][3]));
      }
      *reinterpret_cast<v2u *>(Dh + offd) = (v2u){h0, h1};
      if constexpr (!D16) *reinterpret_cast<v2u *>(Dl + offd) = (v2u){l0, l1};   /* a bf16 operand has no low part */
      if constexpr (A16) {
        unpair(av[p & ~1], p & 1, ap[p] != nullptr, h0, h1); l0 = 0u; l1 = 0u;
      } else {
        split_pair(f(av[p][0]), f(av[p][1]), h0, l0);
        split_pair(f(av[p][2]), f(av[p][3]), h1, l1);
      }
      *reinterpret_cast<v2u *>(Ah + offa) = (v2u){h0, h1};
      if constexpr (!A16) *reinterpret_cast<v2u *>(Al + offa) = (v2u){l0, l1};
    }
    __syncthreads();
    if (k0 + WB_KT < k_end) fetch(k0 + WB_KT);         /* next tile's loads fly under this tile's MFMAs */
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < WB_KT / 16; ++kk) {
      v8bf ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ro = (wm * 64 + i * 32 + sl) * WB_ROW + kk * 32 + h * 16;
        const int co = (wn * 64 + i * 32 + sl) * WB_ROW + kk * 32 + h * 16;
        ah[i] = *reinterpret_cast<const v8bf *>(Dh + ro);
        if constexpr (!D16) al[i] = *reinterpret_cast<const v8bf *>(Dl + ro);
        bh[i] = *reinterpret_cast<const v8bf *>(Ah + co);
        if constexpr (!A16) bl[i] = *reinterpret_cast<const v8bf *>(Al + co);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          /* the low-part products exist only for fp32 operands: 3, 2 or 1 MFMA per product */
          if constexpr (!D16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (!A16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float *part = A.part + (size_t)slice * (EXT ? EXT_PARAMS : NUM_PARAMS);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int colk = tn * WG_TN + wn * 64 + j * 32 + sl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = tm * WG_TM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < J.n_out && colk < J.n_in) part[wjob_row_off(J, orow) + colk] = acc[i][j][r];
      }
    }
  if (tn == 0 && J.b_off >= 0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float s = bsum[p];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      const int orow = tm * WG_TM + tile_row(p, D16);
      if ((tid & 15) == 0 && orow < J.n_out) part[wjob_bias_off(J, orow)] = s;
    }
  }
}

}  // namespace rn
