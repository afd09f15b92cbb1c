/*
 * refnerf_wgrad_f16.h -- the weight-gradient contraction on the split-f16 formats of refnerf_layout.h (round 4):
 *   DELTA = ONE IEEE half per element (delta * c_s, 11 bits; c_s = the power-of-two factor the backward chain carried for
 *           sample s in that layer, kept per (layer id, sample) in the DSC units),
 *   ACT   = hi / lo pair units (22 bits), exactly as the chain kernels held them.
 * dW[o][k] = sum_s DELTA[o][s] ACT[k][s] = (1 / c_min) sum_s (d16[o][s] * c_min / c_s) (a_hi[k][s] + a_lo[k][s]):
 * every sample is brought to the layer's smallest factor c_min (taken over the samples that HAVE a gradient in that layer) with
 * one packed multiply by a power of two (exact; a sample 2^29 below the layer's largest deltas starts to lose bits, 2^39 below
 * it vanishes -- it carries that little of the sum),
 * then TWO v_mfma_f32_32x32x16_f16 per product (d * a_hi + d * a_lo, fp32 accumulate) instead of the three of the split-bf16
 * GEMM, no on-the-fly operand split (its 6 VALU per MFMA), and 26 KB instead of 34.5 KB of operands per ray-sample.
 * Why 11 bits suffice for DELTA: scripts/exp_delta_precision.py (the reference's autograd with rounded weight-gradient
 * operands on the three trained weight sets: gradient rel-L2 2e-5 .. 7e-5; bf16's 8 bits give 2e-4 .. 6e-4).
 * Same job table, split-K slices, PART layout and fixed-order reduction as the other two GEMMs: bit-reproducible, no atomics
 * (the min over the factors is exact in any order).
 * Restates what autograd does for nn.Linear (internal/models.py:576-580,686-700): dW = delta^T x, db = sum delta.
 */
#pragma once
#include "refnerf_wgrad_bf16x3.h"

namespace rn {

/* c_min[lid] = min over the valid samples of the factor row of layer id `lid`; cmin[] pre-set to +inf bits.
 * grid = (DSC_ROWS, 64) x 256 threads; positive floats order like their bit patterns, so one atomicMin per block. */
__global__ __launch_bounds__(256) void delta_scale_min(const float *__restrict__ delta, long long S, float *cmin) {
  const int lid = blockIdx.x;
  const float *row = delta + (long long)(DSC0 + lid) * RB;
  float m = INFINITY;
  for (long long s = (long long)blockIdx.y * blockDim.x + threadIdx.x; s < S; s += (long long)gridDim.y * blockDim.x) {
    const float c = row[rb_col(s, DEL_UNITS_F16S)];
    m = (c > 0.0f) ? fminf(m, c) : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m < INFINITY) atomicMin(reinterpret_cast<int *>(cmin) + lid, __builtin_bit_cast(int, m));
}

typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_mul_h(unsigned a, unsigned b) {
  const v2h z = __builtin_bit_cast(v2h, a) * __builtin_bit_cast(v2h, b);
  return __builtin_bit_cast(unsigned, z);
}
__device__ __forceinline__ float pk_sum_h(unsigned a) {
  const v2h z = __builtin_bit_cast(v2h, a);
  const _Float16 z0 = z[0], z1 = z[1];
  return (float)z0 + (float)z1;
}
/* LDS tiles of this kernel: 128 rows x 64 halves = 128 B per row, NO pad -- the 16-B chunk c of row r sits at chunk
 * c ^ ((r >> 1) & 7): the sixteen lanes ds_read_b128 serves per cycle (consecutive rows, one chunk) then cover the sixteen
 * 16-B slots of the 256-B bank window exactly once, as the 144-B pitch of the bf16 GEMM does -- and three tiles are 48 KB
 * instead of 54 (bit-identical results, same time).  Three workgroups per CU then fit the LDS: measured below. */
constexpr int WF_ROW = WB_KT * 2;              /* 128 B */
__device__ __forceinline__ int wf_off(int row, int byte) { return row * WF_ROW + ((((byte >> 4) ^ (row >> 1)) & 7) << 4) + (byte & 15); }
#ifndef REFNERF_WF_OCC
#define REFNERF_WF_OCC 2                       /* waves per SIMD the register allocation aims at (4-wave workgroups: workgroups per CU) */
#endif
/* the backward scales a sample's largest delta into [2^7, 2^8) (pow2_scale_for); the sample(s) with the layer's smallest factor
 * go up another 2^7 on load, to just below the largest half (2^15 < 65504), everything else follows: 29 binades at full
 * precision below the layer's largest deltas, 10 more of gradual underflow */
constexpr float TOP_SHIFT = 128.0f;

/* grid = 8 * ceil(slices / 8) * tiles workgroups of 64 NW threads, decoded as in wgrad_bf16x3_kernel.
 * Default: NW = 4 waves, D (delta) tile of TM = 128 rows: waves 2 x 2 of 64 x 64 over a 128 x 128 output tile (job table WJOBS),
 * 48 KB of LDS, 172 registers, two workgroups per CU: 3.6 ms per level at C2 = 20 GB actually fetched (13.7 GB of operands: the
 * 2 x 2 tiles of a layer re-read either operand, L2 catches a third of that) at 5.5 TB/s.
 * Measured alternatives (round 4, C2; bit-identical results, docs/EXPERIMENTS.md section 9):
 *   TM = 256 (ALL output rows of a layer per tile, job table WJOBS_M256: ACT -- the larger operand -- is read once):
 *     NW = 8 (waves 4 x 2 of 64 x 64), one workgroup per CU: 17.4 GB fetched, 3.80 ms; with two register sets of loads
 *     (REFNERF_WF_STAGES8 = 2): 15.5 GB, 4.16 ms; two workgroups per CU need <= 128 registers: 120 B/lane of scratch, 6.5 ms;
 *     NW = 4 (waves 2 x 2 of 128 x 64): 256 registers + 64 B/lane, 5.2 ms;
 *   TM = 128, three workgroups per CU (REFNERF_WF_OCC = 3: 168 registers + 16 B/lane): 3.76 ms.
 * Fewer bytes at lower occupancy, or more workgroups at the same bytes, both lose: the GEMM sits at the HBM rate of what it
 * fetches, and fetching less needs the 256-row tile at two workgroups per CU -- 64 accumulator + 32 load registers leave no room. */
#ifndef REFNERF_WF_WAVES
#define REFNERF_WF_WAVES 4
#endif
#ifndef REFNERF_WF_STAGES8
#define REFNERF_WF_STAGES8 1
#endif
constexpr bool wjobs_rows_even(const WJobs &T) {
  for (int j = 0; j < T.n; ++j) if ((T.job[j].a_row | T.job[j].d_row) & 1) return false;
  return true;
}
static_assert(wjobs_rows_even(WJOBS) && wjobs_rows_even(WJOBS_M256), "pair units: every job starts on an even row of ACT and DELTA");
constexpr int WF_NW = REFNERF_WF_WAVES;
#ifndef REFNERF_WF_TM
#define REFNERF_WF_TM 128
#endif
constexpr int wf_tm(int nw) { return REFNERF_WF_TM; }                 /* rows of the D tile: 128, or 256 = all output rows of a layer */
constexpr int wf_lds(int nw) { return (wf_tm(nw) + 2 * WG_TN) * WF_ROW; }
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(REFNERF_WF_OCC))) void wgrad_f16s_kernel(const WgradArgs A, int slices, const float *__restrict__ cmin_all) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int TM = wf_tm(NW);                  /* 128 or 256 */
  constexpr int MI = TM / (16 * NW);             /* 32-row blocks of the D tile per wave: waves (NW / 2) x 2 of (32 MI) x 64 */
  constexpr int NPD = TM / (8 * NW);             /* row PAIRS of the D tile per loader thread: 4 */
  constexpr int NPA = WG_TN / (8 * NW);          /* ... of the A tile: 4 or 2 */
  constexpr int NRD = 2 * NPD, NRA = 2 * NPA;    /* rows */
  constexpr const WJobs &JT = (TM == 256) ? WJOBS_M256 : WJOBS;
  extern __shared__ __attribute__((aligned(16))) char wbs[];
  char *Dh = wbs, *Ah = wbs + TM * WF_ROW, *Al = Ah + WG_TN * WF_ROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
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
  const int lid = del_layer_id(J.d_row);
  const float cmin = cmin_all[lid];
  const bool have = cmin < INFINITY;             /* (no valid sample wrote a factor: nothing to add) */
  const bool need_bias = tn == 0 && J.b_off >= 0;

  v16f acc[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[NRD];
#pragma unroll
  for (int p = 0; p < NRD; ++p) bsum[p] = 0.0f;

  /* loader: the thread owns NPD PAIRS of rows of the D tile and NPA of the A tile (p = 2 pp + half, as the bf16 pair format of
   * wgrad_bf16x3_kernel), 4 samples at lc4.  DELTA: one 16-B load per pair; ACT: the pair's hi unit and its lo unit.
   * The pair index is a bit permutation of the low four bits of lrow (two row groups of a half-wave 8 rows apart in LDS). */
  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
  const int lpair = (lrow & ~7) | ((lrow & 1) << 2) | ((lrow >> 1) & 3);
  auto tile_row = [&](int p) { return 2 * lpair + 8 * NW * (p >> 1) + (p & 1); };
  /* ONE address per operand: pair pp of this thread sits 4 NW pair units behind pair pp - 1 (the liveness of a row is a compare) */
  const char *dp0 = reinterpret_cast<const char *>(A.delta) + ((long long)((J.d_row + tm * TM) / 2 + lpair) * RB + lc4) * 4;
  const char *ap0 = reinterpret_cast<const char *>(A.act) + ((long long)(J.a_row + tn * WG_TN + 2 * lpair) * RB + lc4) * 4;
  const long long dpp = (long long)(4 * NW) * RB * 4, app = (long long)(8 * NW) * RB * 4;   /* bytes between a thread's pairs */
  auto d_live = [&](int p) { return tm * TM + tile_row(p) < J.n_out; };
  auto a_live = [&](int p) { return tn * WG_TN + tile_row(p) < J.n_in; };
  const char *scp = reinterpret_cast<const char *>(A.delta) + ((long long)(DSC0 + lid) * RB + lc4) * 4;
  const long long dstep = (long long)A.d_units * 4, astep = (long long)A.a_units * 4;   /* bytes per sample of k0 */
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  /* NST register sets of operand loads in flight (k-steps i, i + 1, ...): ONE (a second set costs the 4-wave form its second
   * workgroup per CU: 7.35 ms; the 8-wave form alone on its CU: 4.16 ms against 3.80) */
  constexpr int NST = (NW == 8) ? REFNERF_WF_STAGES8 : 1;
  v4u dv[NST][NPD], avh[NST][NPA], avl[NST][NPA];
  v4f cv[NST];
  auto fetch = [&](auto SETC, long long k0) {
    constexpr int st = decltype(SETC)::value;
#pragma unroll
    for (int pp = 0; pp < NPD; ++pp) {
      v4u x = {0u, 0u, 0u, 0u};
      if (d_live(2 * pp)) x = *reinterpret_cast<const v4u *>(dp0 + pp * dpp + k0 * dstep);
      dv[st][pp] = x;
    }
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
      v4u y = {0u, 0u, 0u, 0u}, z = {0u, 0u, 0u, 0u};
      if (a_live(2 * pp)) {
        y = *reinterpret_cast<const v4u *>(ap0 + pp * app + k0 * astep);
        z = *reinterpret_cast<const v4u *>(ap0 + pp * app + k0 * astep + RB * 4);
      }
      avh[st][pp] = y; avl[st][pp] = z;
    }
    cv[st] = *reinterpret_cast<const v4f *>(scp + k0 * dstep);
  };
  auto unpair = [](const v4u w, int half, bool live, unsigned &s01, unsigned &s23) {
    const unsigned sel = half ? 0x07060302u : 0x05040100u;
    s01 = live ? __builtin_amdgcn_perm(w[1], w[0], sel) : 0u;
    s23 = live ? __builtin_amdgcn_perm(w[3], w[2], sel) : 0u;
  };
  auto step = [&](auto SETC, long long k0) {
    constexpr int st = decltype(SETC)::value;
    __syncthreads();                                   /* previous tile fully consumed */
    /* this thread's four samples to the layer's smallest factor: c_min / c_s, a power of two <= 1 (select, not multiply:
     * the factor of a pad sample is whatever the allocator left there) */
    unsigned f01, f23;
    {
      float f[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)      /* (factors are powers of two: v_rcp_f32 is exact on them) */
        f[i] = (have && k0 + lc4 + i < A.S && cv[st][i] > 0.0f) ? (cmin * TOP_SHIFT) * __builtin_amdgcn_rcpf(cv[st][i]) : 0.0f;
      f01 = pk_f16(f[0], f[1]);
      f23 = pk_f16(f[2], f[3]);
    }
#pragma unroll
    for (int p = 0; p < NRD; ++p) {
      unsigned h0, h1;
      unpair(dv[st][p >> 1], p & 1, d_live(p), h0, h1);
      h0 = pk_mul_h(h0, f01);
      h1 = pk_mul_h(h1, f23);
      if (need_bias) bsum[p] += pk_sum_h(h0) + pk_sum_h(h1);      /* (wave-uniform: only the first column tile of a job with a bias) */
      *reinterpret_cast<v2u *>(Dh + wf_off(tile_row(p), lc4 * 2)) = (v2u){h0, h1};
    }
#pragma unroll
    for (int p = 0; p < NRA; ++p) {
      const int off = wf_off(tile_row(p), lc4 * 2);
      unsigned h0, h1, l0, l1;
      unpair(avh[st][p >> 1], p & 1, a_live(p), h0, h1);
      unpair(avl[st][p >> 1], p & 1, a_live(p), l0, l1);
      *reinterpret_cast<v2u *>(Ah + off) = (v2u){h0, h1};
      *reinterpret_cast<v2u *>(Al + off) = (v2u){l0, l1};
    }
    __syncthreads();
    if (k0 + NST * WB_KT < k_end) fetch(SETC, k0 + NST * WB_KT);   /* this set's next tile flies under NST tiles of MFMAs */
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < WB_KT / 16; ++kk) {
      v8h dh[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) dh[i] = *reinterpret_cast<const v8h *>(Dh + wf_off(wm * 32 * MI + i * 32 + sl, kk * 32 + h * 16));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = wf_off(wn * 64 + j * 32 + sl, kk * 32 + h * 16);
        const v8h bh = *reinterpret_cast<const v8h *>(Ah + co), bl = *reinterpret_cast<const v8h *>(Al + co);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bh, acc[i][j], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, NST - 1> S1;
  if (k_begin < k_end) fetch(S0(), k_begin);
  if (NST > 1 && k_begin + WB_KT < k_end) fetch(S1(), k_begin + WB_KT);
  for (long long k0 = k_begin; k0 < k_end; k0 += NST * WB_KT) {
    step(S0(), k0);
    if (NST > 1 && k0 + WB_KT < k_end) step(S1(), k0 + WB_KT);
  }
  const float inv = have ? 1.0f / (cmin * TOP_SHIFT) : 0.0f;   /* (a power of two: exact) */
  float *part = A.part + (size_t)slice * NUM_PARAMS;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int colk = tn * WG_TN + wn * 64 + j * 32 + sl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = tm * TM + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < J.n_out && colk < J.n_in) part[wjob_row_off(J, orow) + colk] = acc[i][j][r] * inv;
      }
    }
  if (need_bias) {
#pragma unroll
    for (int p = 0; p < NRD; ++p) {
      float s = bsum[p];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      const int orow = tm * TM + tile_row(p);
      if ((tid & 15) == 0 && orow < J.n_out) part[wjob_bias_off(J, orow)] = s * inv;
    }
  }
}

}  // namespace rn
