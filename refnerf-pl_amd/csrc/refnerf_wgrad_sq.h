/*
 * refnerf_wgrad_sq.h -- the weight-gradient contraction on the round-5 formats (refnerf_sq_layout.h): wgrad_f16s_kernel of
 * refnerf_wgrad_f16.h with
 *   * ACT in REFNERF_ACT_SQ: spatial jobs read hi / lo pair units (two products d * a_hi + d * a_lo), directional jobs read the
 *     ONE half the forward's trunk multiplied (one product, half the ACT bytes and half the MFMAs of those jobs);
 *   * DELTA factors as (c, kappa) units: every sample of a layer is brought to the layer's smallest kappa K (the power of two that
 *     puts the LARGEST delta of the layer at [2^14, 2^15)): d16 * (K / c_s), then the tile is divided by K.  The backward's
 *     bound-based factors leave a sample's stored maximum anywhere in the half's range, so the common factor comes from kappa,
 *     the multiplier from c.
 * Same job geometry, split-K slices, PART layout and fixed-order reduction: bit-reproducible, no atomics.
 * Restates what autograd does for nn.Linear (internal/models.py:576-580,686-700): dW = delta^T x, db = sum delta.
 */
#pragma once
#include "refnerf_wgrad.h"
#include "refnerf_sq_layout.h"

namespace rn {

/* the job table on the units of REFNERF_ACT_SQ: a_row = first UNIT of the job's input, `half` = one half per element */
struct WJobSq { WJob j; int a_unit; int half; };
struct WJobsSq { WJobSq job[MAX_WJOBS]; int n; int tiles; };
constexpr WJobsSq make_wjobs_sq() {
  WJobsSq T{};
  for (int i = 0; i < WJOBS.n; ++i) {
    WJobSq q{};
    q.j = WJOBS.job[i];
    const int a = q.j.a_row;
    if (a < ACT_DIN) { q.a_unit = (a == ACT_IPE) ? AQ_IPE : AQ_SP + (a - ACT_SP); q.half = 0; }
    else if (a == ACT_DIN) { q.a_unit = AQ_DIN; q.half = 1; }
    else { q.a_unit = AQ_VD + (a - ACT_VD) / 2; q.half = 1; }
    T.job[i] = q;
  }
  T.n = WJOBS.n;
  T.tiles = WJOBS.tiles;
  return T;
}
constexpr WJobsSq WJOBS_SQ = make_wjobs_sq();

/* K[lid] = min over the valid samples of the kappa unit of layer id `lid`; kmin[] pre-set to +inf bits.
 * The DSC_ROWS kappa units of a 64-sample block are adjacent (4.5 KB): a wave takes whole blocks, lane = sample, all rows
 * in flight at once (one row per workgroup column walked the matrix eighteen times in 256-byte steps: 62 us at C2).
 * grid = any x 256 threads; positive floats order like their bit patterns: atomicMin on the bits. */
__global__ __launch_bounds__(256) void delta_kappa_min(const float *__restrict__ delta, long long S, float *kmin) {
  const int lane = threadIdx.x & 63;
  const long long nblk = (S + RB - 1) / RB, nw = (long long)gridDim.x * 4;
  float m[DSC_ROWS];
#pragma unroll
  for (int lid = 0; lid < DSC_ROWS; ++lid) m[lid] = INFINITY;
  for (long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); b < nblk; b += nw) {
    const float *p = delta + (b * DQ_UNITS + DQ_K) * RB + lane;
    const bool ok = b * RB + lane < S;
    float c[DSC_ROWS];
#pragma unroll
    for (int lid = 0; lid < DSC_ROWS; ++lid) c[lid] = p[lid * RB];
#pragma unroll
    for (int lid = 0; lid < DSC_ROWS; ++lid) m[lid] = (ok && c[lid] > 0.0f) ? fminf(m[lid], c[lid]) : m[lid];
  }
  /* one atomic per workgroup and row (atomics on one address queue up in the L2: 2048 waves x 18 rows of them took 0.3 ms) */
  __shared__ float wm[4][DSC_ROWS];
#pragma unroll
  for (int lid = 0; lid < DSC_ROWS; ++lid) {
    float v = m[lid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    if (lane == 0) wm[threadIdx.x >> 6][lid] = v;
  }
  __syncthreads();
  if (threadIdx.x < DSC_ROWS) {
    const float v = fminf(fminf(wm[0][threadIdx.x], wm[1][threadIdx.x]), fminf(wm[2][threadIdx.x], wm[3][threadIdx.x]));
    if (v < INFINITY) atomicMin(reinterpret_cast<int *>(kmin) + threadIdx.x, __builtin_bit_cast(int, v));
  }
}

typedef _Float16 sqw_v2h __attribute__((ext_vector_type(2)));
typedef _Float16 sqw_v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned sqw_pk_mul(unsigned a, unsigned b) {
  const sqw_v2h z = __builtin_bit_cast(sqw_v2h, a) * __builtin_bit_cast(sqw_v2h, b);
  return __builtin_bit_cast(unsigned, z);
}
__device__ __forceinline__ float sqw_pk_sum(unsigned a) {
  const sqw_v2h z = __builtin_bit_cast(sqw_v2h, a);
  const _Float16 z0 = z[0], z1 = z[1];
  return (float)z0 + (float)z1;
}
__device__ __forceinline__ unsigned sqw_pk_f16(float lo, float hi) {
  const sqw_v2h r = __builtin_convertvector((v2f){lo, hi}, sqw_v2h);
  return __builtin_bit_cast(unsigned, r);
}
constexpr int SQW_KT = 64;                     /* samples per k-step = one block of the operand matrices */
constexpr int SQW_ROW = SQW_KT * 2;            /* 128 B per LDS row, chunks swizzled as in refnerf_wgrad_f16.h */
__device__ __forceinline__ int sqw_off(int row, int byte) { return row * SQW_ROW + ((((byte >> 4) ^ (row >> 1)) & 7) << 4) + (byte & 15); }
constexpr int SQW_NW = 4, SQW_TM = 128;
constexpr int SQW_LDS = (SQW_TM + 2 * WG_TN) * SQW_ROW;
static_assert(SQW_KT == RB, "one k-step = one 64-sample block of the operand matrices");

struct WgradSqArgs {
  const float *act, *delta;
  long long S;
  int k_per_slice;
  float *part;            /* [slices][NUM_PARAMS] */
};

/* grid = 8 * ceil(slices / 8) * tiles workgroups of 256 threads: waves 2 x 2 of 64 x 64 over a 128 x 128 output tile */
/* act11 (cfg.wgrad_mode = REFNERF_WGRAD_F16): the spatial jobs take the hi units of their pair units only (the forward did not
 * write the lo units): one product per tile everywhere */
__global__ __launch_bounds__(64 * SQW_NW) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_sq_kernel(const WgradSqArgs A, int slices, const float *__restrict__ kmin_all, int act11) {
  constexpr int NW = SQW_NW, TM = SQW_TM;
  constexpr int MI = TM / (16 * NW);             /* 2 */
  constexpr int NPD = TM / (8 * NW);             /* row PAIRS of the D tile per loader thread: 4 */
  constexpr int NPA = WG_TN / (8 * NW);          /* ... of the A tile: 4 */
  constexpr int NRD = 2 * NPD, NRA = 2 * NPA;
  extern __shared__ __attribute__((aligned(16))) char wbs[];
  char *Dh = wbs, *Ah = wbs + TM * SQW_ROW, *Al = Ah + WG_TN * SQW_ROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile = q % WJOBS_SQ.tiles, slice = (q / WJOBS_SQ.tiles) * 8 + xcd;
  if (slice >= slices) return;
  int ji = 0;
#pragma unroll 1
  for (int j = 1; j < WJOBS_SQ.n; ++j) if (tile >= WJOBS_SQ.job[j].j.tile0) ji = j;
  const WJob J = WJOBS_SQ.job[ji].j;
  const int a_unit = WJOBS_SQ.job[ji].a_unit;
  const bool halfrows = WJOBS_SQ.job[ji].half != 0;          /* ONE half per element, rows in pairs: one unit per pair row */
  const bool half = halfrows || act11 != 0;                  /* no lo operand */
  const int tl = tile - J.tile0;
  const int tm = tl / J.tiles_n, tn = tl - tm * J.tiles_n;
  const long long k_begin = (long long)slice * A.k_per_slice;
  long long k_end = k_begin + A.k_per_slice;
  const long long s_pad = (A.S + SQW_KT - 1) / SQW_KT * SQW_KT;
  if (k_end > s_pad) k_end = s_pad;
  const int lid = del_layer_id(J.d_row);
  const float kmin = kmin_all[lid];
  const bool have = kmin < INFINITY;
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

  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
  const int lpair = (lrow & ~7) | ((lrow & 1) << 2) | ((lrow >> 1) & 3);
  auto tile_row = [&](int p) { return 2 * lpair + 8 * NW * (p >> 1) + (p & 1); };
  /* pair pp of this thread sits 4 NW pair rows behind pair pp - 1: DELTA one unit per pair row, ACT two (hi, lo) or one */
  const int aup = halfrows ? 1 : 2;                                 /* ACT units per pair row */
  const char *dp0 = reinterpret_cast<const char *>(A.delta) + ((long long)((J.d_row + tm * TM) / 2 + lpair) * RB + lc4) * 4;
  const char *ap0 = reinterpret_cast<const char *>(A.act) + ((long long)(a_unit + (tn * (WG_TN / 2) + lpair) * aup) * RB + lc4) * 4;
  const long long dpp = (long long)(4 * NW) * RB * 4, app = (long long)(4 * NW * aup) * RB * 4;
  auto d_live = [&](int p) { return tm * TM + tile_row(p) < J.n_out; };
  auto a_live = [&](int p) { return tn * WG_TN + tile_row(p) < J.n_in; };
  const char *scp = reinterpret_cast<const char *>(A.delta) + ((long long)(DQ_C + lid) * RB + lc4) * 4;
  const long long dstep = (long long)DQ_UNITS * 4, astep = (long long)AQ_UNITS * 4;   /* bytes per sample of k0 */
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  v4u dv[NPD], avh[NPA], avl[NPA];
  v4f cv;
  auto fetch = [&](long long k0) {
#pragma unroll
    for (int pp = 0; pp < NPD; ++pp) {
      v4u x = {0u, 0u, 0u, 0u};
      if (d_live(2 * pp)) x = *reinterpret_cast<const v4u *>(dp0 + pp * dpp + k0 * dstep);
      dv[pp] = x;
    }
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
      v4u y = {0u, 0u, 0u, 0u}, z = {0u, 0u, 0u, 0u};
      if (a_live(2 * pp)) {
        y = *reinterpret_cast<const v4u *>(ap0 + pp * app + k0 * astep);
        if (!half) z = *reinterpret_cast<const v4u *>(ap0 + pp * app + k0 * astep + RB * 4);
      }
      avh[pp] = y; avl[pp] = z;
    }
    cv = *reinterpret_cast<const v4f *>(scp + k0 * dstep);
  };
  auto unpair = [](const v4u w, int hf, bool live, unsigned &s01, unsigned &s23) {
    const unsigned sel = hf ? 0x07060302u : 0x05040100u;
    s01 = live ? __builtin_amdgcn_perm(w[1], w[0], sel) : 0u;
    s23 = live ? __builtin_amdgcn_perm(w[3], w[2], sel) : 0u;
  };
  auto step = [&](long long k0) {
    __syncthreads();                                   /* previous tile fully consumed */
    /* this thread's four samples to the layer's common factor: K / c_s, a power of two (factors are powers of two: v_rcp_f32 is
     * exact on them; a pad sample's factor is whatever the allocator left there: select, not multiply) */
    /* (K / c_s = 2^(15 - e(stored max)) x (K / kappa_s) passes the largest half only for a sample whose stored deltas are all
     * below 1: applied as two powers of two then) */
    unsigned f01, f23, g01, g23;
    {
      float f[4], g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = (have && k0 + lc4 + i < A.S && cv[i] > 0.0f) ? kmin * __builtin_amdgcn_rcpf(cv[i]) : 0.0f;
        f[i] = fminf(x, 32768.0f);
        g[i] = x > 32768.0f ? x * (1.0f / 32768.0f) : 1.0f;
      }
      f01 = sqw_pk_f16(f[0], f[1]);
      f23 = sqw_pk_f16(f[2], f[3]);
      g01 = sqw_pk_f16(g[0], g[1]);
      g23 = sqw_pk_f16(g[2], g[3]);
    }
#pragma unroll
    for (int p = 0; p < NRD; ++p) {
      unsigned h0, h1;
      unpair(dv[p >> 1], p & 1, d_live(p), h0, h1);
      h0 = sqw_pk_mul(sqw_pk_mul(h0, f01), g01);
      h1 = sqw_pk_mul(sqw_pk_mul(h1, f23), g23);
      if (need_bias) bsum[p] += sqw_pk_sum(h0) + sqw_pk_sum(h1);
      *reinterpret_cast<v2u *>(Dh + sqw_off(tile_row(p), lc4 * 2)) = (v2u){h0, h1};
    }
#pragma unroll
    for (int p = 0; p < NRA; ++p) {
      const int off = sqw_off(tile_row(p), lc4 * 2);
      unsigned h0, h1, l0, l1;
      unpair(avh[p >> 1], p & 1, a_live(p), h0, h1);
      *reinterpret_cast<v2u *>(Ah + off) = (v2u){h0, h1};
      if (!half) {
        unpair(avl[p >> 1], p & 1, a_live(p), l0, l1);
        *reinterpret_cast<v2u *>(Al + off) = (v2u){l0, l1};
      }
    }
    __syncthreads();
    if (k0 + SQW_KT < k_end) fetch(k0 + SQW_KT);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < SQW_KT / 16; ++kk) {
      sqw_v8h dh[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) dh[i] = *reinterpret_cast<const sqw_v8h *>(Dh + sqw_off(wm * 32 * MI + i * 32 + sl, kk * 32 + h * 16));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = sqw_off(wn * 64 + j * 32 + sl, kk * 32 + h * 16);
        const sqw_v8h bh = *reinterpret_cast<const sqw_v8h *>(Ah + co);
        if (!half) {
          const sqw_v8h bl = *reinterpret_cast<const sqw_v8h *>(Al + co);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bl, acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bh, acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  if (k_begin < k_end) fetch(k_begin);
  for (long long k0 = k_begin; k0 < k_end; k0 += SQW_KT) step(k0);
  const float inv = have ? 1.0f / kmin : 0.0f;   /* (a power of two: exact) */
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

/* ---------------------------------------------------------------------------------------------------------------------------
 * The 256 x 256 tile: ONE workgroup of eight waves owns a whole layer's dW for its slice of the sample axis, so every DELTA and
 * ACT byte of the slice is fetched exactly once (the 128 x 128 tiles above read each operand twice and the four sibling tiles
 * drift apart faster than one XCD's L2 can hold the 3 MB a k-step of its 64 workgroups streams: 19 % hits, 18.9 GB fetched
 * for 11.6 GB of operands at 4096 x 128).  Waves 4 x 2 of 64 x 128 (128 accumulator registers); 32-sample k-steps.
 *
 * One workgroup per CU has nobody to hide its HBM latency behind, and 128 accumulators + fragments leave no room for a second
 * register set of operands in flight (two sets: 280-1300 spilled registers in every arrangement tried, also as four waves
 * with 512 registers each).  So the operands travel HBM -> LDS by LDS-DMA into a ring of RAW k-steps -- every lane DMAs the 7 x
 * 16 B it will itself convert, so a ring slot is a per-lane FIFO and needs no barrier, only the wave's own vmcnt -- and two
 * k-steps (2 x 56 KB per CU) are in flight while a third is converted (unpair, common factor, transpose) into the fragment
 * layout T and multiplied.  LDS: T 48 KB + 2 x 56 KB = 160 KB (no lo operand: T 32 KB + 3 x 40 KB).  Jobs issued heaviest first.
 * Same arithmetic and slice order as wgrad_sq_kernel (the bias sums add in another order).
 * (Round 6: the shipped kernel runs wgrad_sq256_raw_body below -- no conversion pass, no T; this body builds with
 * -DREFNERF_SQ2_CONVERT_PASS for A/B runs.)
 * ------------------------------------------------------------------------------------------------------------------------- */
constexpr int SQ2_KT = 32, SQ2_T = 256, SQ2_ROWB = SQ2_KT * 2;
constexpr int SQ2_REG = SQ2_T * SQ2_ROWB;                      /* one operand of T: 16 KB */
constexpr int SQ2_TB = 3 * SQ2_REG;                            /* T = [D | A_hi | A_lo] */
constexpr int SQ2_RG = 512 * 16;                               /* one 16 B piece per thread: 8 KB */
/* ring slot = D x 2, A_hi x 2, (A_lo x 2,) factors; two slots beside the three operands of T, or -- no lo operand -- three slots
 * beside two */
constexpr int sq2_regions(bool half) { return half ? 5 : 7; }
constexpr int sq2_slots(bool half) { return half ? 3 : 2; }
constexpr int sq2_tbytes(bool half) { return (half ? 2 : 3) * SQ2_REG; }
constexpr int sq2_lds(bool half) { return sq2_tbytes(half) + sq2_slots(half) * sq2_regions(half) * SQ2_RG; }
constexpr int SQ2_LDS = sq2_lds(false) > sq2_lds(true) ? sq2_lds(false) : sq2_lds(true);
static_assert(SQ2_LDS <= 160 * 1024, "LDS of one CU");
static_assert(WIDTH <= SQ2_T && DIR_IN <= SQ2_T && IPE_DIM <= SQ2_T && HROWS <= SQ2_T, "one tile per job");
struct Sq2Order { int o[MAX_WJOBS]; };
constexpr Sq2Order make_sq2_order() {
  Sq2Order O{};
  int cost[MAX_WJOBS] = {};
  for (int i = 0; i < WJOBS_SQ.n; ++i) {
    O.o[i] = i;
    cost[i] = ((WJOBS_SQ.job[i].j.n_out + 1) / 2) * 4 + WJOBS_SQ.job[i].j.n_in * (WJOBS_SQ.job[i].half ? 2 : 4);
  }
  for (int i = 1; i < WJOBS_SQ.n; ++i)
    for (int k = i; k > 0 && cost[O.o[k]] > cost[O.o[k - 1]]; --k) { const int t = O.o[k]; O.o[k] = O.o[k - 1]; O.o[k - 1] = t; }
  return O;
}
constexpr Sq2Order SQ2_ORDER = make_sq2_order();
__device__ __forceinline__ long long sq2_koff(long long k, int units) { return ((k >> 6) * (long long)units * RB + (k & 63)) * 4; }
typedef __attribute__((address_space(1))) const void *sq2_gptr;
typedef __attribute__((address_space(3))) void *sq2_lptr;

/* HALF: no lo operand (directional jobs: one half per element; every job under cfg.wgrad_mode = REFNERF_WGRAD_F16) -- a template
 * parameter, so that each flavour is straight-line code */
template <bool HALF>
__device__ __forceinline__ void wgrad_sq256_body(const WgradSqArgs &A, int slice, int ji, const float *__restrict__ kmin_all) {
  extern __shared__ __attribute__((aligned(16))) char wbs[];
  constexpr int VM = sq2_regions(HALF);            /* VMEM operations (all LDS-DMA) of one k-step and wave */
  constexpr int NS = sq2_slots(HALF), SLOT = sq2_regions(HALF) * SQ2_RG, TB = sq2_tbytes(HALF);
  constexpr int G_AH = 2, G_AL = 4, G_CV = HALF ? 4 : 6;      /* region numbers within a slot */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const WJob J = WJOBS_SQ.job[ji].j;
  const int a_unit = WJOBS_SQ.job[ji].a_unit;
  const bool halfrows = WJOBS_SQ.job[ji].half != 0;
  const long long k_begin = (long long)slice * A.k_per_slice;
  long long k_end = k_begin + A.k_per_slice;
  const long long s_pad = (A.S + RB - 1) / RB * RB;
  if (k_end > s_pad) k_end = s_pad;
  const int nsteps = (int)((k_end - k_begin + SQ2_KT - 1) / SQ2_KT);    /* >= 2: the caller's slice count leaves no slice empty */
  const int lid = del_layer_id(J.d_row);
  const float kmin = kmin_all[lid];
  const bool have = kmin < INFINITY;

  v16f acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  /* this lane's pieces: pair rows p0 = tid / 8 and p0 + 64 of both tiles, samples c4 .. c4 + 3 of the k-step.  A pair row past
   * the job's last one re-reads the last one (same cache lines; the conversion zeroes it) */
  const int p0 = tid >> 3, c4 = (tid & 7) * 4;
  const int aup = halfrows ? 1 : 2;
  const int dlast = (J.n_out - 1) / 2, alast = (J.n_in - 1) / 2;
  const char *dp[2], *ap[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pd = min(p0 + 64 * q, dlast), pa = min(p0 + 64 * q, alast);
    dp[q] = reinterpret_cast<const char *>(A.delta) + ((long long)(J.d_row / 2 + pd) * RB + c4) * 4;
    ap[q] = reinterpret_cast<const char *>(A.act) + ((long long)(a_unit + pa * aup) * RB + c4) * 4;
  }
  const char *scp = reinterpret_cast<const char *>(A.delta) + ((long long)(DQ_C + lid) * RB + c4) * 4;
  auto kof = [&](int s) { return k_begin + (long long)(s < nsteps ? s : nsteps - 1) * SQ2_KT; };   /* (past the end: the last one again) */
  /* k-step s -> ring slot: VM LDS-DMA instructions per wave, lane l of wave w lands at region + (64 w + l) 16 = region + 16 tid */
  auto issue = [&](int s, int slot) {
    const long long k = kof(s);
    const long long dk = sq2_koff(k, DQ_UNITS), ak = sq2_koff(k, AQ_UNITS);
    char *base = wbs + TB + slot * SLOT + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __builtin_amdgcn_global_load_lds((sq2_gptr)(dp[q] + dk), (sq2_lptr)(base + q * SQ2_RG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((sq2_gptr)(ap[q] + ak), (sq2_lptr)(base + (G_AH + q) * SQ2_RG), 16, 0, 0);
      if (!HALF) __builtin_amdgcn_global_load_lds((sq2_gptr)(ap[q] + ak + RB * 4), (sq2_lptr)(base + (G_AL + q) * SQ2_RG), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((sq2_gptr)(scp + dk), (sq2_lptr)(base + G_CV * SQ2_RG), 16, 0, 0);
  };
  auto unpair = [](const v4u w, int hf, bool live, unsigned &s01, unsigned &s23) {
    const unsigned sel = hf ? 0x07060302u : 0x05040100u;
    s01 = live ? __builtin_amdgcn_perm(w[1], w[0], sel) : 0u;
    s23 = live ? __builtin_amdgcn_perm(w[3], w[2], sel) : 0u;
  };
  /* T: 64 B rows of four 16 B chunks, chunk ^ (row / 4) -- the 16 rows a quarter wave reads at one chunk sit on 16 bank groups.
   * Every address = a lane register + an immediate (the swizzle term depends on the lane and on kk only); the registers are
   * made opaque once per k-step, or LICM parks the precomputed sums in VGPRs across the loop */
  const int xs = (sl >> 2) & 3;
  int rdD[2], rdA[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = (((kk << 1) | h) ^ xs) << 4;
    rdD[kk] = (wm * 64 + sl) * SQ2_ROWB + ch;
    rdA[kk] = SQ2_REG + (wn * 128 + sl) * SQ2_ROWB + ch;
  }
  int wrT = (2 * p0) * SQ2_ROWB + ((((c4 >> 3) ^ (p0 >> 1)) & 3) << 4) + ((c4 * 2) & 15);
  int mine = TB + tid * 16;
  auto opaque = [&] { asm volatile("" : "+v"(rdD[0]), "+v"(rdD[1]), "+v"(rdA[0]), "+v"(rdA[1]), "+v"(wrT), "+v"(mine)); };
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  /* k-step s (arrived in `slot`) -> T; the slot is re-armed with k-step s + SQ2_NS as soon as its pieces sit in registers */
  auto stage = [&](int s, int slot) {
    const long long k = kof(s);
    const bool fresh = s < nsteps;             /* (past the end the last k-step comes again: not into the bias sums) */
    const char *src = wbs + mine + slot * SLOT;
    v4u dv[2], avh[2], avl[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      dv[q] = *reinterpret_cast<const v4u *>(src + q * SQ2_RG);
      avh[q] = *reinterpret_cast<const v4u *>(src + (G_AH + q) * SQ2_RG);
      if (!HALF) avl[q] = *reinterpret_cast<const v4u *>(src + (G_AL + q) * SQ2_RG);
    }
    const v4f cv = *reinterpret_cast<const v4f *>(src + G_CV * SQ2_RG);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef REFNERF_EXPERIMENT_SQ2_NODMA
    issue(s + NS, slot);
#endif
#ifdef REFNERF_EXPERIMENT_SQ2_NOSTAGE
    if (dv[0][0] == 0x12345678u && avh[1][2] == 77u && avl[0][1] == 3u && cv[2] == 1.5f) bsum[0] += 1.0f;
    return;
#endif
    unsigned f01, f23, g01, g23;
    {
      float f[4], g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        /* (selects, not branches.  K / c_s, both powers of two: v_rcp_f32 is exact on them; passed as two factors when above the
         * half's range; a pad sample's c is whatever the allocator left there) */
        const float r = kmin * __builtin_amdgcn_rcpf(cv[i]);
        const bool ok = have & (k + c4 + i < A.S) & (cv[i] > 0.0f);
        const float x = ok ? r : 0.0f;
        f[i] = fminf(x, 32768.0f);
        g[i] = x > 32768.0f ? x * (1.0f / 32768.0f) : 1.0f;
      }
      f01 = sqw_pk_f16(f[0], f[1]);
      f23 = sqw_pk_f16(f[2], f[3]);
      g01 = sqw_pk_f16(g[0], g[1]);
      g23 = sqw_pk_f16(g[2], g[3]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = 2 * (p0 + 64 * q) + e;
        const int imm = (128 * q + e) * SQ2_ROWB;
        unsigned h0, h1;
        unpair(dv[q], e, row < J.n_out, h0, h1);
        h0 = sqw_pk_mul(sqw_pk_mul(h0, f01), g01);
        h1 = sqw_pk_mul(sqw_pk_mul(h1, f23), g23);
        bsum[2 * q + e] += fresh ? sqw_pk_sum(h0) + sqw_pk_sum(h1) : 0.0f;   /* (a select: a branch would split the k-step into blocks) */
        *reinterpret_cast<v2u *>(wbs + wrT + imm) = (v2u){h0, h1};
        unpair(avh[q], e, row < J.n_in, h0, h1);
        *reinterpret_cast<v2u *>(wbs + wrT + SQ2_REG + imm) = (v2u){h0, h1};
        if (!HALF) {
          unpair(avl[q], e, row < J.n_in, h0, h1);
          *reinterpret_cast<v2u *>(wbs + wrT + 2 * SQ2_REG + imm) = (v2u){h0, h1};
        }
      }
  };
  auto compute = [&] {
#ifdef REFNERF_EXPERIMENT_SQ2_NOCOMPUTE
    return;
#endif
#pragma unroll
    for (int kk = 0; kk < SQ2_KT / 16; ++kk) {
      sqw_v8h dh[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) dh[i] = *reinterpret_cast<const sqw_v8h *>(wbs + rdD[kk] + i * 32 * SQ2_ROWB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const sqw_v8h bh = *reinterpret_cast<const sqw_v8h *>(wbs + rdA[kk] + j * 32 * SQ2_ROWB);
        if (!HALF) {
          const sqw_v8h bl = *reinterpret_cast<const sqw_v8h *>(wbs + rdA[kk] + SQ2_REG + j * 32 * SQ2_ROWB);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bl, acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[i], bh, acc[i][j], 0, 0, 0);
      }
    }
  };
  /* this wave's VM oldest DMA instructions (= the k-step about to be converted) have landed; the younger k-step stays in flight */
#ifdef REFNERF_EXPERIMENT_SQ2_NODMA
  auto arrived = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#else
  auto arrived = [] { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM * (NS - 1)) : "memory"); };
#endif
  auto t_written = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto t_consumed = [] { asm volatile("s_barrier" ::: "memory"); };   /* (a wave's T reads are behind its MFMAs: done) */
#pragma unroll
  for (int s = 0; s < NS; ++s) issue(s, s);
  arrived();
  stage(0, 0);
  t_written();
  int slot = 1;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    opaque();
    compute();
    t_consumed();
    arrived();
    stage(s + 1, slot);              /* (past the last k-step: the last one again, into a T nobody reads) */
    slot = slot + 1 == NS ? 0 : slot + 1;
    t_written();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* (the re-armed slots' DMA before the workgroup gives its LDS back) */
  const bool need_bias = J.b_off >= 0;
  const float inv = have ? 1.0f / kmin : 0.0f;
  float *part = A.part + (size_t)slice * NUM_PARAMS;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int colk = wn * 128 + j * 32 + sl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < J.n_out && colk < J.n_in) part[wjob_row_off(J, orow) + colk] = acc[i][j][r] * inv;
      }
    }
  if (need_bias) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float sum = bsum[p];
      sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
      const int orow = 2 * (p0 + 64 * (p >> 1)) + (p & 1);
      if ((tid & 7) == 0 && orow < J.n_out) part[wjob_bias_off(J, orow)] = sum * inv;
    }
  }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Round 6: the one-half jobs WITHOUT the conversion pass (wgrad_sq256_raw_body).  The ring holds the operands as they lie in
 * HBM -- pair units: one dword = rows 2p, 2p + 1 of one sample -- and every wave turns pair units into MFMA operands when it
 * loads its fragments: a lane's eight dwords of pair row p are the k-halves of rows 2p AND 2p + 1, so the even rows of 32 pair
 * rows are one 32-row operand block and the odd rows the next (v_perm_b32; the common factor K / c_s as the halves (f_2t,
 * f_2t+1) multiplies the operand dword of samples 2t, 2t + 1).  No T, no second barrier, no LDS round trip of the converted
 * tile, and the LDS that T took is a fourth ring slot.
 * Ring slot = [D: 128 pair rows x 128 B | A: 128 pair rows x 128 B | c: 8 waves x 128 B]; 16-B chunk c of pair row p at chunk
 * c ^ ((p >> 1) & 7) (the DMA lane picks its global chunk accordingly: the swizzle is free).  Rows are 128 B and the LDS has 64
 * banks: the 16 rows a quarter wave reads at one chunk are 8 row PAIRS, each pair the two halves of one 256-B bank line -- so the
 * chunk is swizzled by the pair's number (swizzled by the row's own number, rows r and r + 8 met on the same banks: 44 % of the
 * kernel's LDS cycles were bank conflicts, profiles/r06 of the first build).
 * What bounds the compute side is the SIMD's one issue port: a k-step is 16 MFMAs (32 cycles each) + ~100 VALU instructions
 * per wave, two waves per SIMD in lockstep (one barrier per k-step).  Measured on the way (docs/EXPERIMENTS.md section 11): the
 * conversion at fragment-load time but phase by phase (loads, wait, convert, 8 MFMAs) 2.17 -> 1.89 ms; the same with the loads
 * half a k-step ahead but the MFMAs still back to back: 2.04 (VALU time and MFMA time simply add); every MFMA followed by the
 * 4 - 8 VALU instructions that fit under it, the D fragment of the NEXT half k-step among them: 1.75; scalar DMA bases, the
 * second factor on a scalar branch, bias sums on one wave per SIMD only: 1.71 ms = 0.67 of 8 TB/s (DMA alone: 1.63).
 * The 22-bit mode's spatial jobs (LO: a third operand, three 49 KB slots, four units of eight MFMAs per k-step): 3.19 -> 2.75 ms.
 * ------------------------------------------------------------------------------------------------------------------------- */
constexpr int SQ3_NS = 4;
#ifndef REFNERF_SQ3_DMA_AUX
#define REFNERF_SQ3_DMA_AUX 2      /* cache policy bits of the operand stream's LDS-DMA (sc0 = 1, nt = 2, sc1 = 16): every byte is read
                                    * once, by one CU -- nt; measured 0.5-1 % against the default policy, sc1 none (EXPERIMENTS section 11) */
#endif
#ifndef REFNERF_SQ3_SPLIT
#define REFNERF_SQ3_SPLIT 32768.0f     /* (a smaller power of two sends every k-step through the two-factor path: a test build, same arithmetic) */
#endif
constexpr float SQ3_SPLIT = REFNERF_SQ3_SPLIT;
constexpr int SQ3_OP = 128 * 128;                       /* one operand of a k-step: 128 pair rows x 32 samples x 4 B */
constexpr int SQ3_SLOT = 2 * SQ3_OP + 8 * 128;
constexpr int SQ3_FS = SQ3_NS * SQ3_SLOT;               /* per wave: two buffers of a k-step's factors as halves, f and g: 2 x 2 x 64 B */
constexpr int SQ3_LDS = SQ3_FS + 8 * 256;
static_assert(SQ3_NS == 4 && SQ3_LDS <= SQ2_LDS, "inside the kernel's LDS");
typedef _Float16 sqw_h2 __attribute__((ext_vector_type(2)));

#ifdef REFNERF_EXPERIMENT_SQ3_NOMFMA
#define SQ3_MFMA(a, b, c, x, y, z) ({ asm volatile("" ::"v"(a), "v"(b)); (c); })
#else
#define SQ3_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#endif
/* this wave's part of the k-step that is due has landed (KSTEPS younger ones may still fly), then everybody's */
#define SQ3_WAIT_BARRIER(KSTEPS) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM * (KSTEPS)) : "memory")
/* BIAS: this wave carries the bias sums of its 64 rows (the waves of column half 0 of a job with a bias: one per SIMD) */
template <bool V> struct Sq3Tag { static constexpr bool value = V; };
/* LO: the 22-bit mode's spatial jobs -- a third operand, the lo halves of the layer inputs (pair units one unit behind their hi
 * units): slot = D | A_hi | A_lo | c = 49 KB, three slots, and a k-step is FOUR units of eight MFMAs (kk 0 hi, kk 0 lo, kk 1 hi,
 * kk 1 lo): each unit's A fragments are loaded during the unit before, the D fragment of the next kk is converted during the lo unit */
template <bool BIAS, bool LO>
__device__ __forceinline__ void wgrad_sq256_raw_body(const WgradSqArgs &A, int slice, int ji, const float *__restrict__ kmin_all) {
  extern __shared__ __attribute__((aligned(16))) char wbs[];
  constexpr int NS = LO ? 3 : SQ3_NS, VM = LO ? 7 : 5;
  constexpr int NOP = LO ? 3 : 2;                     /* operands of a slot */
  constexpr int SLOT = NOP * SQ3_OP + 8 * 128, FSB = NS * SLOT;
  static_assert(FSB + 8 * 256 <= SQ2_LDS, "inside the kernel's LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, sl = lane & 31;
  const int wm = wave & 3, wn = wave >> 2;      /* (waves w and w + 4 share a SIMD: one of each column half) */
  const WJob J = WJOBS_SQ.job[ji].j;
  const int a_unit = WJOBS_SQ.job[ji].a_unit;
  const bool halfrows = WJOBS_SQ.job[ji].half != 0;
  const long long k_begin = (long long)slice * A.k_per_slice;
  long long k_end = k_begin + A.k_per_slice;
  const long long s_pad = (A.S + RB - 1) / RB * RB;
  if (k_end > s_pad) k_end = s_pad;
  const int nsteps = (int)((k_end - k_begin + SQ2_KT - 1) / SQ2_KT);
  const int lid = del_layer_id(J.d_row);
  const float kmin = kmin_all[lid];
  const bool have = kmin < INFINITY;

  v16f acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};

  /* DMA: lane -> pair row 64 q + tid / 8 of both operands, global chunk (tid & 7) ^ ((row >> 1) & 7), LDS chunk tid & 7 */
  const int p0 = tid >> 3, c4 = (((tid & 7) ^ (p0 >> 1)) & 7) * 4;
  const int aup = halfrows ? 1 : 2;
  const int dlast = (J.n_out - 1) / 2, alast = (J.n_in - 1) / 2;
  /* address = a scalar base (matrix + the k-step's place in it: the same for every lane, SALU work) + a lane offset that
   * never changes (32 bits: the units of one 64-sample block span < 1 MB) */
  unsigned doff[2], aoff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pd = min(p0 + 64 * q, dlast), pa = min(p0 + 64 * q, alast);
    doff[q] = (unsigned)(((J.d_row / 2 + pd) * RB + c4) * 4);
    aoff[q] = (unsigned)(((a_unit + pa * aup) * RB + c4) * 4);
  }
  const unsigned coff = (unsigned)(((DQ_C + lid) * RB + (lane & 7) * 4) * 4);
  const char *dmat = reinterpret_cast<const char *>(A.delta), *amat = reinterpret_cast<const char *>(A.act);
  const int kh0 = __builtin_amdgcn_readfirstlane((int)(k_begin >> 5)), nst = __builtin_amdgcn_readfirstlane(nsteps);
  auto issue = [&](int s, int slot) {
    const int kh = kh0 + (s < nst ? s : nst - 1);                    /* k / 32: 64-sample block kh / 2, half kh & 1 */
    const char *db = dmat + ((long long)(kh >> 1) * (DQ_UNITS * RB * 4) + (kh & 1) * 128);
    const char *ab = amat + ((long long)(kh >> 1) * (AQ_UNITS * RB * 4) + (kh & 1) * 128);
    char *base = wbs + slot * SLOT + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __builtin_amdgcn_global_load_lds((sq2_gptr)(db + doff[q]), (sq2_lptr)(base + q * 8192), 16, 0, REFNERF_SQ3_DMA_AUX);
      __builtin_amdgcn_global_load_lds((sq2_gptr)(ab + aoff[q]), (sq2_lptr)(base + SQ3_OP + q * 8192), 16, 0, REFNERF_SQ3_DMA_AUX);
      if (LO) __builtin_amdgcn_global_load_lds((sq2_gptr)(ab + aoff[q] + RB * 4), (sq2_lptr)(base + 2 * SQ3_OP + q * 8192), 16, 0, REFNERF_SQ3_DMA_AUX);
    }
    if (lane < 8) __builtin_amdgcn_global_load_lds((sq2_gptr)(db + coff), (sq2_lptr)(wbs + slot * SLOT + NOP * SQ3_OP + wave * 128), 16, 0, REFNERF_SQ3_DMA_AUX);
  };
  /* fragment addresses: pair row (wm 32 + sl) of D, (wn 64 + jj 32 + sl) of A; logical chunk 4 kk + 2 h + e at ^ ((sl >> 1) & 7) */
  const int x7 = (sl >> 1) & 7;
  int choff[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int e = 0; e < 2; ++e) choff[kk][e] = (((4 * kk + 2 * h + e) ^ x7) & 7) << 4;
  int rowD = (wm * 32 + sl) * 128, rowA = SQ3_OP + (wn * 64 + sl) * 128;
  /* per wave, two buffers (k-step parity) of the k-step's factors as halves: f of sample k at + 2 k, g at + 64 + 2 k */
  int fsl = FSB + wave * 256 + h * 16, fsw = FSB + wave * 256 + sl * 2, cvr = NOP * SQ3_OP + wave * 128 + sl * 4;
  const unsigned one2 = 0x3c003c00u;
  typedef unsigned v4uu __attribute__((ext_vector_type(4)));
  auto frag = [](const v4uu w0, const v4uu w1, unsigned sel) {
    v4uu r;
    r[0] = __builtin_amdgcn_perm(w0[1], w0[0], sel);
    r[1] = __builtin_amdgcn_perm(w0[3], w0[2], sel);
    r[2] = __builtin_amdgcn_perm(w1[1], w1[0], sel);
    r[3] = __builtin_amdgcn_perm(w1[3], w1[2], sel);
    return r;
  };
  /* one half k-step (16 samples) of this wave's fragments as they lie in the ring: 6 x 16 B per lane */
  struct Half { v4uu d0, d1, a[2][2]; };
  struct Fac { v4uu f, g; };
  auto load_half = [&](int so, int kk) {
    Half H;
    H.d0 = *reinterpret_cast<const v4uu *>(wbs + so + rowD + choff[kk][0]);
    H.d1 = *reinterpret_cast<const v4uu *>(wbs + so + rowD + choff[kk][1]);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 2; ++e) H.a[jj][e] = *reinterpret_cast<const v4uu *>(wbs + so + rowA + jj * 4096 + choff[kk][e]);
    return H;
  };
  auto load_fac = [&](int buf, int kk, bool with_g) {
    Fac F;
    F.f = *reinterpret_cast<const v4uu *>(wbs + fsl + buf * 128 + kk * 32);
    F.g = F.f;
    if (with_g) F.g = *reinterpret_cast<const v4uu *>(wbs + fsl + buf * 128 + kk * 32 + 64);
    return F;
  };
  /* the k-step's 32 factors K / c_s, one per lane (both halves of the wave the same): f = min(x, 2^15) and -- a sample whose
   * stored deltas are all below 1 -- g = x / 2^15, as halves; an operand dword after the unpair holds samples 2 t, 2 t + 1 of
   * one row, so it takes the halves (f_2t, f_2t+1) as they lie */
  auto put_factors = [&](float cvv, int s, int buf) {
    const long long left = A.S - (k_begin + (long long)s * SQ2_KT);
    const int lim = left < 0 ? 0 : (left > 32 ? 32 : (int)left);
    const bool ok = have & (s < nst) & (sl < lim) & (cvv > 0.0f);   /* (s = nst: the loop's look-ahead past its last k-step -- factor 0) */
    const float x = ok ? kmin * __builtin_amdgcn_rcpf(cvv) : 0.0f;
    const float f = fminf(x, SQ3_SPLIT);
    const bool big = x > SQ3_SPLIT;
    const bool slow = __builtin_amdgcn_ballot_w64(big) != 0;          /* (seldom: the second factor costs 16 instructions per k-step) */
    *reinterpret_cast<_Float16 *>(wbs + fsw + buf * 128) = (_Float16)f;
    if (slow) *reinterpret_cast<_Float16 *>(wbs + fsw + buf * 128 + 64) = (_Float16)(big ? x * (1.0f / SQ3_SPLIT) : 1.0f);
    return slow;
  };
  /* Hand-ordered half k-steps.  The two waves of a SIMD run in lockstep (one barrier per k-step), so a wave that issues its eight
   * MFMAs back to back leaves the VALU port idle for 256 cycles and then both waves queue on it: measured, the VALU work and
   * the MFMAs of a k-step simply ADD (1.41 ms of compute = 0.93 + 0.48).  Here every MFMA is followed by the 4 - 8 VALU
   * instructions that fit under it: first the unpair of this half's own A fragments, then -- their loads issued at the top of
   * the half -- the D fragment of the NEXT half (unpair, common factor, bias sums).  sched_barrier(0) after every group: the
   * source order is the issue order. */
  auto conv_d_part = [&](bool slow, const Half &H, const Fac &F, int t, v4uu &e4, v4uu &o4) {
    const unsigned w0 = t < 2 ? H.d0[2 * t] : H.d1[2 * t - 4], w1 = t < 2 ? H.d0[2 * t + 1] : H.d1[2 * t - 3];
    const unsigned ft = F.f[t], gt = F.g[t];
    unsigned e = __builtin_amdgcn_perm(w1, w0, 0x05040100u), o = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
    e = sqw_pk_mul(e, ft);
    o = sqw_pk_mul(o, ft);
    if (__builtin_expect(slow, 0)) {      /* (a real branch on the scalar flag: the empty asm keeps it from becoming two selects) */
      asm volatile("");
      e = sqw_pk_mul(e, gt); o = sqw_pk_mul(o, gt);
    }
    if (BIAS) {
      bsum[0] = __builtin_amdgcn_fdot2(__builtin_bit_cast(sqw_h2, e), __builtin_bit_cast(sqw_h2, one2), bsum[0], false);
      bsum[1] = __builtin_amdgcn_fdot2(__builtin_bit_cast(sqw_h2, o), __builtin_bit_cast(sqw_h2, one2), bsum[1], false);
    }
    e4[t] = e; o4[t] = o;
  };
#define SQ3_SB() __builtin_amdgcn_sched_barrier(0)
  /* MFMAs of the half whose operands are (dE, dO, H.a) + the D fragment of the half (Hn, Fn) -> (nE, nO) */
  /* (so_next >= 0: (Hn, Fn) = the second half of the k-step in slot so_next, loaded here BEHIND the first MFMA -- the first
   * unpair waits for lgkmcnt(0), which must not include loads issued a moment ago) */
  auto half_step = [&](bool slow, const v4uu dE4, const v4uu dO4, const Half &H, Half &Hn, Fac &Fn, v4uu &nE, v4uu &nO, int so_next, int buf) {
    const sqw_v8h dE = __builtin_bit_cast(sqw_v8h, dE4), dO = __builtin_bit_cast(sqw_v8h, dO4);
    const sqw_v8h aE0 = __builtin_bit_cast(sqw_v8h, frag(H.a[0][0], H.a[0][1], 0x05040100u));
    SQ3_SB();
    acc[0][0] = SQ3_MFMA(dE, aE0, acc[0][0], 0, 0, 0);
    if (so_next >= 0) { Hn = load_half(so_next, 1); Fn = load_fac(buf, 1, slow); }
    const sqw_v8h aO0 = __builtin_bit_cast(sqw_v8h, frag(H.a[0][0], H.a[0][1], 0x07060302u));
    SQ3_SB();
    acc[1][0] = SQ3_MFMA(dO, aE0, acc[1][0], 0, 0, 0);
    const sqw_v8h aE1 = __builtin_bit_cast(sqw_v8h, frag(H.a[1][0], H.a[1][1], 0x05040100u));
    SQ3_SB();
    acc[0][1] = SQ3_MFMA(dE, aO0, acc[0][1], 0, 0, 0);
    const sqw_v8h aO1 = __builtin_bit_cast(sqw_v8h, frag(H.a[1][0], H.a[1][1], 0x07060302u));
    SQ3_SB();
    acc[1][1] = SQ3_MFMA(dO, aO0, acc[1][1], 0, 0, 0);
    conv_d_part(slow, Hn, Fn, 0, nE, nO);
    SQ3_SB();
    acc[0][2] = SQ3_MFMA(dE, aE1, acc[0][2], 0, 0, 0);
    conv_d_part(slow, Hn, Fn, 1, nE, nO);
    SQ3_SB();
    acc[1][2] = SQ3_MFMA(dO, aE1, acc[1][2], 0, 0, 0);
    conv_d_part(slow, Hn, Fn, 2, nE, nO);
    SQ3_SB();
    acc[0][3] = SQ3_MFMA(dE, aO1, acc[0][3], 0, 0, 0);
    conv_d_part(slow, Hn, Fn, 3, nE, nO);
    SQ3_SB();
    acc[1][3] = SQ3_MFMA(dO, aO1, acc[1][3], 0, 0, 0);
    asm volatile("" : "+v"(nE), "+v"(nO));      /* (used here: or MachineSink carries the conversion off to the block of its first MFMA, behind the barrier) */
    SQ3_SB();
  };
  auto opaque = [&] { asm volatile("" : "+v"(rowD), "+v"(rowA), "+v"(fsl), "+v"(fsw), "+v"(cvr), "+v"(choff[0][0]), "+v"(choff[0][1]), "+v"(choff[1][0]), "+v"(choff[1][1])); };
  if constexpr (LO) {
    /* ---- four units per k-step ---- */
    struct A4 { v4uu a[2][2]; };
    struct D2 { v4uu d0, d1; };
    auto load_a = [&](int so, int kk, int op) {          /* op 1 = hi, 2 = lo */
      A4 R;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 2; ++e) R.a[jj][e] = *reinterpret_cast<const v4uu *>(wbs + so + rowA + (op - 1) * SQ3_OP + jj * 4096 + choff[kk][e]);
      return R;
    };
    auto load_d = [&](int so, int kk) {
      D2 R;
      R.d0 = *reinterpret_cast<const v4uu *>(wbs + so + rowD + choff[kk][0]);
      R.d1 = *reinterpret_cast<const v4uu *>(wbs + so + rowD + choff[kk][1]);
      return R;
    };
    auto conv = [&](bool slow, const D2 &Dr, const Fac &F, int t, v4uu &e4, v4uu &o4) {
      Half Hh;
      Hh.d0 = Dr.d0; Hh.d1 = Dr.d1;
      conv_d_part(slow, Hh, F, t, e4, o4);
    };
    /* one unit: eight MFMAs (dE, dO) x the four operand blocks of Ar; `behind0()` = the loads issued behind the first MFMA;
     * CONV: the D fragment (Dn, Fn) -> (nE, nO) behind MFMAs 4 .. 7 */
    auto unit = [&](auto conv_tag, bool slow, const v4uu dE4, const v4uu dO4, const A4 &Ar, auto &&behind0, const D2 &Dn, const Fac &Fn, v4uu &nE, v4uu &nO) {
      constexpr bool CONV = decltype(conv_tag)::value;
      const sqw_v8h dE = __builtin_bit_cast(sqw_v8h, dE4), dO = __builtin_bit_cast(sqw_v8h, dO4);
      const sqw_v8h aE0 = __builtin_bit_cast(sqw_v8h, frag(Ar.a[0][0], Ar.a[0][1], 0x05040100u));
      SQ3_SB();
      acc[0][0] = SQ3_MFMA(dE, aE0, acc[0][0], 0, 0, 0);
      behind0();
      const sqw_v8h aO0 = __builtin_bit_cast(sqw_v8h, frag(Ar.a[0][0], Ar.a[0][1], 0x07060302u));
      SQ3_SB();
      acc[1][0] = SQ3_MFMA(dO, aE0, acc[1][0], 0, 0, 0);
      const sqw_v8h aE1 = __builtin_bit_cast(sqw_v8h, frag(Ar.a[1][0], Ar.a[1][1], 0x05040100u));
      SQ3_SB();
      acc[0][1] = SQ3_MFMA(dE, aO0, acc[0][1], 0, 0, 0);
      const sqw_v8h aO1 = __builtin_bit_cast(sqw_v8h, frag(Ar.a[1][0], Ar.a[1][1], 0x07060302u));
      SQ3_SB();
      acc[1][1] = SQ3_MFMA(dO, aO0, acc[1][1], 0, 0, 0);
      if (CONV) conv(slow, Dn, Fn, 0, nE, nO);
      SQ3_SB();
      acc[0][2] = SQ3_MFMA(dE, aE1, acc[0][2], 0, 0, 0);
      if (CONV) conv(slow, Dn, Fn, 1, nE, nO);
      SQ3_SB();
      acc[1][2] = SQ3_MFMA(dO, aE1, acc[1][2], 0, 0, 0);
      if (CONV) conv(slow, Dn, Fn, 2, nE, nO);
      SQ3_SB();
      acc[0][3] = SQ3_MFMA(dE, aO1, acc[0][3], 0, 0, 0);
      if (CONV) conv(slow, Dn, Fn, 3, nE, nO);
      SQ3_SB();
      acc[1][3] = SQ3_MFMA(dO, aO1, acc[1][3], 0, 0, 0);
      if (CONV) asm volatile("" : "+v"(nE), "+v"(nO));
      SQ3_SB();
    };
    typedef Sq3Tag<true> Yes;
    typedef Sq3Tag<false> No;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    SQ3_WAIT_BARRIER(NS - 2);
    bool slow0 = put_factors(*reinterpret_cast<const float *>(wbs + cvr), 0, 0);
    v4uu dE, dO, nE, nO;
    A4 Ah = load_a(0, 0, 1), Al;
    {
      const D2 D0 = load_d(0, 0);
      const Fac F0 = load_fac(0, 0, slow0);
#pragma unroll
      for (int t = 0; t < 4; ++t) conv(slow0, D0, F0, t, dE, dO);
    }
    int i0 = 0;                                       /* slot of k-step s */
    D2 Dn; Fac Fn;
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      opaque();
      const int b = s & 1, i1 = i0 + 1 == NS ? 0 : i0 + 1, ia = i0 == 0 ? NS - 1 : i0 - 1;
      const int so0 = i0 * SLOT, so1 = i1 * SLOT;
      /* kk 0 hi: the lo fragments of kk 0 on their way */
      unit(No{}, slow0, dE, dO, Ah, [&] { Al = load_a(so0, 0, 2); }, Dn, Fn, nE, nO);
      /* kk 0 lo: D, factors and hi fragments of kk 1 on their way, its D fragment converted */
      unit(Yes{}, slow0, dE, dO, Al, [&] { Dn = load_d(so0, 1); Fn = load_fac(b, 1, slow0); Ah = load_a(so0, 1, 1); }, Dn, Fn, nE, nO);
      /* kk 1 hi */
      unit(No{}, slow0, nE, nO, Ah, [&] { Al = load_a(so0, 1, 2); }, Dn, Fn, dE, dO);
      SQ3_WAIT_BARRIER(NS - 3);
      const float cvv = *reinterpret_cast<const float *>(wbs + so1 + cvr);
      SQ3_SB();
      issue(s + NS - 1, ia);
      SQ3_SB();
      const bool slow1 = put_factors(cvv, s + 1, b ^ 1);
      SQ3_SB();
      Dn = load_d(so1, 0);
      Fn = load_fac(b ^ 1, 0, slow1);
      Ah = load_a(so1, 0, 1);
      SQ3_SB();
      /* kk 1 lo: the next k-step's first D fragment converted */
      unit(Yes{}, slow1, nE, nO, Al, [] {}, Dn, Fn, dE, dO);
      slow0 = slow1;
      i0 = i1;
    }
  } else {
  /* The loop runs half a k-step ahead of its MFMAs.  The barrier sits in the MIDDLE of k-step s: there every wave has its part
     * of k-step s + 1 (own vmcnt) and has read the last of k-step s - 1, whose slot is re-armed with k-step s + 3. */
  #pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    SQ3_WAIT_BARRIER(NS - 2);
    bool slow0 = put_factors(*reinterpret_cast<const float *>(wbs + cvr), 0, 0);
    Half R0 = load_half(0, 0);
    v4uu dE, dO, nE, nO;
    {
      const Fac F0 = load_fac(0, 0, slow0);
  #pragma unroll
      for (int t = 0; t < 4; ++t) conv_d_part(slow0, R0, F0, t, dE, dO);
    }
  #pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      opaque();
      const int b = s & 1, so0 = (s & 3) * SLOT, so1 = ((s + 1) & 3) * SLOT;     /* (NS = 4) */
      Half R1;
      Fac F1;
  #ifndef REFNERF_EXPERIMENT_SQ3_NOCOMPUTE
      half_step(slow0, dE, dO, R0, R1, F1, nE, nO, so0, b);
  #endif
      SQ3_WAIT_BARRIER(NS - 3);
      const float cvv = *reinterpret_cast<const float *>(wbs + so1 + cvr);     /* (on its way while the DMA addresses are made) */
      SQ3_SB();
  #ifndef REFNERF_EXPERIMENT_SQ3_NODMA
      issue(s + NS - 1, (s + NS - 1) & 3);
  #endif
      SQ3_SB();
      slow0 = put_factors(cvv, s + 1, b ^ 1);
      SQ3_SB();
      R0 = load_half(so1, 0);
      Fac F0 = load_fac(b ^ 1, 0, slow0);
      SQ3_SB();
  #ifndef REFNERF_EXPERIMENT_SQ3_NOCOMPUTE
      half_step(slow0, nE, nO, R1, R0, F0, dE, dO, -1, 0);
  #endif
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* (the re-armed slots' DMA before the workgroup gives its LDS back) */
  const float inv = have ? 1.0f / kmin : 0.0f;
  float *part = A.part + (size_t)slice * NUM_PARAMS;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int colk = wn * 128 + (j >> 1) * 64 + 2 * sl + (j & 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
        if (orow < J.n_out && colk < J.n_in) part[wjob_row_off(J, orow) + colk] = acc[i][j][r] * inv;
      }
    }
  if (BIAS) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float sum = bsum[i];
      sum += __shfl_xor(sum, 32, 64);
      const int orow = wm * 64 + 2 * sl + i;
      if (h == 0 && orow < J.n_out) part[wjob_bias_off(J, orow)] = sum * inv;
    }
  }
}

/* grid = jobs x slices workgroups of 512 threads, the heaviest jobs first.  (Slices per job in proportion to the job's bytes --
 * two even rounds of ~495 workgroups -- measured SLOWER, 3.9 against 3.3 ms: a k-step of a light job costs the same MFMA and
 * conversion time as a heavy one's, so its few long workgroups became the tail: docs/EXPERIMENTS.md section 10.) */
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_sq256_kernel(const WgradSqArgs A, int slices, const float *__restrict__ kmin_all, int act11) {
#ifdef REFNERF_EXPERIMENT_SQ2_SLICE_MAJOR
  const int slice = blockIdx.x / WJOBS_SQ.n, tix = blockIdx.x - slice * WJOBS_SQ.n;
#else
  const int tix = blockIdx.x / slices, slice = blockIdx.x - tix * slices;
#endif
  const int ji = SQ2_ORDER.o[tix];
  const bool half = WJOBS_SQ.job[ji].half != 0 || act11 != 0;      /* no lo operand */
#ifdef REFNERF_SQ2_CONVERT_PASS                                    /* round 5's bodies with their conversion pass (A/B builds) */
  if (half) wgrad_sq256_body<true>(A, slice, ji, kmin_all);
  else wgrad_sq256_body<false>(A, slice, ji, kmin_all);
#else
  const bool bias = threadIdx.x < 256 && WJOBS_SQ.job[ji].j.b_off >= 0;     /* (waves 0-3 = column half 0) */
  if (half) {
    if (bias) wgrad_sq256_raw_body<true, false>(A, slice, ji, kmin_all);
    else wgrad_sq256_raw_body<false, false>(A, slice, ji, kmin_all);
  } else {
    if (bias) wgrad_sq256_raw_body<true, true>(A, slice, ji, kmin_all);
    else wgrad_sq256_raw_body<false, true>(A, slice, ji, kmin_all);
  }
#endif
}

}  // namespace rn
