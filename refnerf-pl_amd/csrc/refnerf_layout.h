/*
 * refnerf_layout.h -- compile-time layouts shared by the pack kernel, the level
 * kernels and the host side of librefnerf_hip.so.
 *
 * (1) canonical parameter blob = the reference's state_dict order for
 *     nerf_mlp.* (internal/models.py:497-531), row-major [out][in];
 * (2) fp32 MFMA operand image ("packed weights") streamed by the level kernel.
 */
#pragma once

namespace rn {

constexpr int WIDTH = 256;
constexpr int DEPTH = 8;
constexpr int IPE_DIM = 96;
constexpr int BNECK = 128;
constexpr int IDE_TERMS = 36;
constexpr int IDE_DIM = 72;
constexpr int DIR_IN = 201;   /* bottleneck 128 | IDE 72 | n.v 1            */
constexpr int DIR_PAD = 204;  /* padded: 102 K=2 steps, a multiple of the prefetch depth */
constexpr int NUM_PARAMS = 1110158;

/* ---------------- canonical blob ---------------- */
struct Canon {
  int sp_w[DEPTH], sp_b[DEPTH], sp_in[DEPTH];
  int density_w, density_b, gradpred_w, gradpred_b, rough_w, rough_b;
  int diffuse_w, diffuse_b, tint_w, tint_b, bneck_w, bneck_b;
  int vd_w[DEPTH], vd_b[DEPTH], vd_in[DEPTH];
  int rgb_w, rgb_b, total;
};

constexpr Canon make_canon() {
  Canon c{};
  int p = 0;
  for (int i = 0; i < DEPTH; ++i) {
    int in = (i == 0) ? IPE_DIM : (i == 5 ? WIDTH + IPE_DIM : WIDTH);
    c.sp_in[i] = in; c.sp_w[i] = p; p += WIDTH * in; c.sp_b[i] = p; p += WIDTH;
  }
  c.density_w = p; p += WIDTH; c.density_b = p; p += 1;
  c.gradpred_w = p; p += 3 * WIDTH; c.gradpred_b = p; p += 3;
  c.rough_w = p; p += WIDTH; c.rough_b = p; p += 1;
  c.diffuse_w = p; p += 3 * WIDTH; c.diffuse_b = p; p += 3;
  c.tint_w = p; p += 3 * WIDTH; c.tint_b = p; p += 3;
  c.bneck_w = p; p += BNECK * WIDTH; c.bneck_b = p; p += BNECK;
  for (int i = 0; i < DEPTH; ++i) {
    int in = (i == 0) ? DIR_IN : (i == 5 ? WIDTH + DIR_IN : WIDTH);
    c.vd_in[i] = in; c.vd_w[i] = p; p += WIDTH * in; c.vd_b[i] = p; p += WIDTH;
  }
  c.rgb_w = p; p += 3 * WIDTH; c.rgb_b = p; p += 3;
  c.total = p;
  return c;
}
constexpr Canon CANON = make_canon();
static_assert(CANON.total == NUM_PARAMS, "canonical layout");

/* ---------------- fp32 MFMA operand image ----------------
 * 18 GEMM ops per sample block.  Each op computes D[out][sample] =
 * W[out][k] * X[k][sample] with v_mfma_f32_32x32x2_f32 (A = W, B = X), so a
 * lane (sample = lane&31, half h = lane>>5) ends up holding output rows
 * row(reg,h) = (reg&3) + 8*(reg>>2) + 4*h of every 32-row block -- exactly the
 * (k = row) values the next op needs as its B operand for the K=2 step
 * (block kb, reg r).  The weight image is therefore stored per step as
 *   A[step][lane][ob] = W[32*ob + (lane&31)][ kidx(step, lane>>5) ]
 * (8-block ops: stored as two planes [step][q][lane][4 blocks], q = ob / 4, so that each dwordx4 load of the wave is
 * one contiguous KB; 3- / 1-block ops: [step][lane][stride])
 * "register" steps (kb,r): kidx = 32*kb + (r&3) + 8*(r>>2) + 4*h,
 * "LDS" steps s (encoded inputs staged in LDS): kidx = 2*s + h.
 * Bias image: B[ob][h][reg] = bias[32*ob + row(reg,h)] (accumulator seed). */
constexpr int NUM_OPS = 18;
constexpr int OP_HEADS = 8;
constexpr int OP_RGB = 17;
constexpr int REG_STEPS = 128;   /* 256 inputs / 2 */

/* Transposed ("backward-data") ops, same operand format with A = W^T:
 * D[in_row][sample] = sum_o W[o][in_row] * delta[o][sample]; K = the layer's 256
 * outputs (all register steps), rows = the layer's inputs in blocks of 32.
 *   TOP_SP(i)  i = 1..7 : spatial layer i, input rows 0..255        (8 blocks)
 *   TOP_SP5_IPE / TOP_SP0: the 96 IPE inputs of layers 5 / 0        (3 blocks)
 *   TOP_VD(i)  i = 1..7 : directional layer i, input rows 0..255    (8 blocks)
 *   TOP_VD5_DIN / TOP_VD0: the 201 dir-encoding inputs of layers 5/0 (7 blocks, rows >= 201 zero)
 *   TOP_HEADS : the 139 head rows transposed, K = head rows read from LDS
 *               (72 LDS steps, rows >= 139 zero), output rows = the 256 features
 * (used by the density-gradient normals VJP of the training forward and by the
 * backward kernel).  No bias: accumulators start at zero.
 * WD / WRGB: raw_density.weight[256] and rgb_layer.weight[3][256] in
 * accumulator layout (seeds of the two backward chains). */
constexpr int NUM_TOPS = 19;
constexpr int TOP_SP5_IPE = 7, TOP_SP0 = 8, TOP_VD1 = 9, TOP_VD5_DIN = 16, TOP_VD0 = 17, TOP_HEADS = 18;
constexpr int HEADS_T_STEPS = 72;           /* 144 padded head rows / 2 */
constexpr int DIN_BLOCKS = 7;               /* 201 dir inputs -> 224 rows */

struct Op { int nob; int stride; int reg_steps; int lds_k; int lds_steps; int a_off; int b_off; };
struct TopSrc { int fwd_op; int col0; };    /* forward op whose weight is transposed, first input column */
/* bf16 copies of the transposed ops for the bf16-chain backward (cfg.precision = REFNERF_PREC_BF16 in
 * refnerf_level_backward): A fragments of v_mfma_f32_32x32x16_bf16, [k-step][ob (stride 8)][lane][8 bf16], i.e. one
 * 1 KB-contiguous load per (k-step, output block), k order = the accumulator order of the
 * producing MFMA (BT_CHAIN_STEPS = 16 steps: slot (t, h, e) = unit 32 (t >> 1) + row(8 (t & 1) + e, h)); the head
 * block is read from the fp32 LDS tile in plain order (BT_HEADS_STEPS = 9: k = 16 t + 8 h + e, rows >= 139 zero).
 * Offsets in floats inside the same image. */
constexpr int BT_CHAIN_STEPS = 16, BT_HEADS_STEPS = 9, BT_STEP_FLOATS = 64 * 8 * 4;
/* ... and of the 18 forward ops for the bf16-chain training forward (bf_off): register steps as above (16, or 0 for
 * the two layers fed from LDS only), then the LDS-fed inputs in plain order k' = 16 s + 8 h + e over the fp32 LDS tile:
 * BF_IPE_STEPS = 6 (96 IPE features) / BF_DIN_STEPS = 13 (the 204-row dir input, rows >= 201 zero). */
constexpr int BF_IPE_STEPS = 6, BF_DIN_STEPS = 13;
/* ... and split-f16 copies of both sets for the split-chain training forward (cfg.training with REFNERF_PREC_F16X2): per
 * k-step TWO fragment blocks, [k-step][hi | lo][ob][lane][8 halves], w = hi + lo (hf_off: forward ops, ht_off: transposed
 * ops; same step counts and k order as bf_off / bt_off) */
struct Packed { Op op[NUM_OPS]; Op top[NUM_TOPS]; TopSrc top_src[NUM_TOPS]; int wd_off; int wrgb_off; int bt_off[NUM_TOPS];
                int bf_off[NUM_OPS]; int ht_off[NUM_TOPS]; int hf_off[NUM_OPS]; int total; };
constexpr int bf_lds_steps(int op) { return (op == 0 || op == 5) ? BF_IPE_STEPS : ((op == 9 || op == 14) ? BF_DIN_STEPS : 0); }
constexpr int bf_reg_steps(int op) { return (op == 0 || op == 9) ? 0 : BT_CHAIN_STEPS; }

constexpr Packed make_packed() {
  Packed P{};
  int p = 0;
  for (int i = 0; i < NUM_OPS; ++i) {
    Op o{};
    o.nob = (i == OP_HEADS) ? 5 : (i == OP_RGB ? 1 : 8);
    o.stride = (i == OP_RGB) ? 1 : 8;
    bool has_reg = !(i == 0 || i == 9);
    o.reg_steps = has_reg ? REG_STEPS : 0;
    o.lds_k = (i == 0 || i == 5) ? IPE_DIM : ((i == 9 || i == 14) ? DIR_PAD : 0);
    o.lds_steps = o.lds_k / 2;
    o.a_off = p; p += (o.reg_steps + o.lds_steps) * 64 * o.stride;
    o.b_off = p; p += o.nob * 32;
    p = (p + 3) & ~3;
    P.op[i] = o;
  }
  for (int i = 0; i < NUM_TOPS; ++i) {
    Op o{};
    TopSrc t{};
    if (i < 7) { o.nob = 8; t.fwd_op = i + 1; t.col0 = 0; }
    else if (i == TOP_SP5_IPE) { o.nob = 3; t.fwd_op = 5; t.col0 = WIDTH; }
    else if (i == TOP_SP0) { o.nob = 3; t.fwd_op = 0; t.col0 = 0; }
    else if (i < TOP_VD5_DIN) { o.nob = 8; t.fwd_op = 9 + (i - TOP_VD1 + 1); t.col0 = 0; }
    else if (i == TOP_VD5_DIN) { o.nob = DIN_BLOCKS; t.fwd_op = 14; t.col0 = WIDTH; }
    else if (i == TOP_VD0) { o.nob = DIN_BLOCKS; t.fwd_op = 9; t.col0 = 0; }
    else { o.nob = 8; t.fwd_op = OP_HEADS; t.col0 = 0; }
    o.stride = (o.nob == 3) ? 4 : 8;
    o.reg_steps = (i == TOP_HEADS) ? 0 : REG_STEPS;
    o.lds_steps = (i == TOP_HEADS) ? HEADS_T_STEPS : 0;
    o.lds_k = 2 * o.lds_steps;
    o.a_off = p; p += (o.reg_steps + o.lds_steps) * 64 * o.stride;
    o.b_off = -1;
    P.top[i] = o;
    P.top_src[i] = t;
  }
  P.wd_off = p; p += 8 * 32;
  P.wrgb_off = p; p += 3 * 8 * 32;
  p += 8 * 64 * 8;            /* tail pad: the fp32 A prefetch runs PF (<= 8) steps past an op */
  p = (p + 3) & ~3;
  for (int i = 0; i < NUM_TOPS; ++i) {
    P.bt_off[i] = p;
    p += ((i == TOP_HEADS) ? BT_HEADS_STEPS : BT_CHAIN_STEPS) * BT_STEP_FLOATS;
  }
  for (int i = 0; i < NUM_OPS; ++i) {
    P.bf_off[i] = p;
    p += (bf_reg_steps(i) + bf_lds_steps(i)) * BT_STEP_FLOATS;
  }
  p += 4 * BT_STEP_FLOATS;            /* tail pad: the bf16 A prefetch runs up to 4 steps past an op */
  for (int i = 0; i < NUM_TOPS; ++i) {
    P.ht_off[i] = p;
    p += ((i == TOP_HEADS) ? BT_HEADS_STEPS : BT_CHAIN_STEPS) * 2 * BT_STEP_FLOATS;
  }
  for (int i = 0; i < NUM_OPS; ++i) {
    P.hf_off[i] = p;
    p += (bf_reg_steps(i) + bf_lds_steps(i)) * 2 * BT_STEP_FLOATS;
  }
  P.total = p + 4 * 2 * BT_STEP_FLOATS;   /* tail pad: the split A prefetch runs up to 2 steps past an op */
  return P;
}
constexpr Packed PACKED = make_packed();

/* ---------------- general IPE bases (NerfMLP.basis_shape / basis_subdivisions; internal/models.py:384-385,482-484) --------
 * The fused kernels are built around the 3 directions of the octahedron/1 basis (96 IPE features).  A basis of 3 G
 * directions (G <= 7: icosahedron/2 = the reference's constructor default has 21) is served as G GROUPS of three
 * directions: group 0 takes the places of the built-in basis (the canonical blob's IPE columns, ops 0 / 5, the X tile),
 * groups 1..G-1 are run through the same 96-row X tile one after the other, each with its own [256][96] weight block of
 * layers 0 and 5 accumulating into the layer's output.  Feature k of a group = 48 (cos block) + 3 (degree) + direction in
 * group, i.e. the canonical order with the group's directions.
 *   parameters: the canonical blob + a tail  W_ext[layer L = 0 (layer 0), 1 (layer 5)][256 rows][(g - 1) 96 + k]  (NUM_PARAMS_EXT;
 *               row-major with 576 columns, so that the tail's weight gradient is one GEMM job per layer)
 *   fp32 image:  PACKED.total floats + [basis 64][12 forward group ops, [48 steps][2][64][4]][12 transposed, [128][64][4]]
 *                + the split-f16 copies of both sets (6 MB in all) */
constexpr int IPE_MAX_GROUPS = 7;
constexpr int EXT_GROUPS = IPE_MAX_GROUPS - 1;
constexpr int EXT_K = EXT_GROUPS * IPE_DIM;                  /* 576 tail columns per row */
constexpr int EXT_PARAMS = 2 * WIDTH * EXT_K;
constexpr int NUM_PARAMS_EXT = NUM_PARAMS + EXT_PARAMS;
constexpr int ext_w_off(int L, int g) { return NUM_PARAMS + L * WIDTH * EXT_K + (g - 1) * IPE_DIM; }   /* + row * EXT_K + k */
constexpr int PEXT_BASIS = PACKED.total;
constexpr int PEXT_FWD = PEXT_BASIS + 64;
constexpr int PEXT_FWD_FLOATS = (IPE_DIM / 2) * 64 * 8;
constexpr int PEXT_T = PEXT_FWD + 2 * EXT_GROUPS * PEXT_FWD_FLOATS;
constexpr int PEXT_T_FLOATS = REG_STEPS * 64 * 4;
/* ... and their split-f16 copies for the split chains (formats of hf_off / ht_off: [k-step][hi | lo][ob (8)][lane][8 halves];
 * forward: the 6 LDS k-steps of op 0, transposed: 16 k-steps with 3 live blocks) */
constexpr int PEXT_HF = PEXT_T + 2 * EXT_GROUPS * PEXT_T_FLOATS + 8 * 64 * 8;       /* (behind the fp32 prefetch pad) */
constexpr int PEXT_HF_FLOATS = BF_IPE_STEPS * 2 * BT_STEP_FLOATS;
constexpr int PEXT_HT = PEXT_HF + 2 * EXT_GROUPS * PEXT_HF_FLOATS;
constexpr int PEXT_HT_FLOATS = BT_CHAIN_STEPS * 2 * BT_STEP_FLOATS;
constexpr int PACKED_EXT_TOTAL = PEXT_HT + 2 * EXT_GROUPS * PEXT_HT_FLOATS + 4 * 2 * BT_STEP_FLOATS;
constexpr int pext_fwd_off(int L, int g) { return PEXT_FWD + (L * EXT_GROUPS + (g - 1)) * PEXT_FWD_FLOATS; }
constexpr int pext_t_off(int L, int g) { return PEXT_T + (L * EXT_GROUPS + (g - 1)) * PEXT_T_FLOATS; }
constexpr int pext_hf_off(int L, int g) { return PEXT_HF + (L * EXT_GROUPS + (g - 1)) * PEXT_HF_FLOATS; }
constexpr int pext_ht_off(int L, int g) { return PEXT_HT + (L * EXT_GROUPS + (g - 1)) * PEXT_HT_FLOATS; }
/* training: the IPE features of groups 1..G-1 (the weight-gradient operand of the tail) are a second blocked matrix
 * behind ACT in the activations buffer: row (g - 1) 96 + k, ACT_EXT_UNITS units per 64-sample block (odd, as ACT's) */
constexpr int ACT_EXT_ROWS = EXT_K, ACT_EXT_UNITS = EXT_K + 1;

/* ---------------- backward workspace (weight-gradient operands) ----------------
 * The backward kernel writes, for every sample s, the input of every linear
 * layer (ACT) and the gradient w.r.t. its pre-activation output (DELTA) as
 * [feature row][sample] fp32 matrices with row pitch `pitch` floats; the
 * weight-gradient kernel then forms dW[o][k] = sum_s DELTA[o][s] * ACT[k][s].
 * Row maps (all row counts multiples of 4): */
constexpr int ACT_IPE = 0;                         /* 96: IPE features                     */
constexpr int ACT_SP = ACT_IPE + IPE_DIM;          /* 8 x 256: spatial activations x0..x7  */
constexpr int ACT_DIN = ACT_SP + 8 * WIDTH;        /* 204: [bottleneck | IDE | n.v | 0 0 0] */
constexpr int ACT_VD = ACT_DIN + DIR_PAD;          /* 8 x 256: directional activations     */
constexpr int ACT_ROWS = ACT_VD + 8 * WIDTH;       /* 4396 */
/* behind the 4396 operand rows: the ReLU sign patterns of the 16 hidden layers as bit masks, so that the
 * backward re-creates a mask from 4 dwords per lane instead of re-reading the 128 activations:
 * row ACT_MASK + 8*layer + 4*h + q, column = sample, holds the dword mk[q] of half-wave h
 * (bit 16*(ob&1) + r of mk[ob>>1] = unit 32*ob + row(r,h) active); layers 0-7 spatial, 8-15 directional */
constexpr int ACT_MASK = ACT_ROWS;
constexpr int ACT_ROWS_TOTAL = ACT_MASK + 16 * 8;  /* 4524 */
/* bf16 format (REFNERF_ACT_BF16): the rows are bf16 pairs and fill float-rows [0, ACT_ROWS / 2); what the backward reads
 * back PER SAMPLE -- x7, v7 (the packed B fragments of the heads / rgb GEMMs) and the 16 layers' ReLU masks -- is kept a
 * second time as a sample-major block behind them: 48 16-byte slots per lane (x7: 0..15, v7: 16..31, masks: 32 + layer),
 * laid out [32-sample chunk][slot][h][sample in chunk][16 B] so that every store / load instruction of a wave is one
 * contiguous 1 KB.  Row-wise reads of those 320 rows (each row 2 MB from the next: a page and a DRAM row per 128 B)
 * were 0.15 of the backward.  The mask rows stay unused in this format. */
constexpr int SMB_SLOTS = 48, SMB_X7 = 0, SMB_V7 = 16, SMB_MASK = 32;
constexpr int SMB_ROW0 = ACT_ROWS / 2 + 3;         /* first float-row of the block: 2201 (odd: it is also the block stride of the bf16 rows) */
static_assert((long long)(ACT_MASK - SMB_ROW0) * 4 >= SMB_SLOTS * 2 * 16 + 64, "sample-major block must fit in front of the mask rows");
/* Storage order of both matrices: BLOCKS OF 64 SAMPLES, [block][unit][64 samples] with one dword per unit and sample
 * (unit = an fp32 row, or a pair-row of the bf16 format): element (unit u, sample s) is dword
 *     (s >> 6) * units * 64 + u * 64 + (s & 63)
 * i.e. the row-major expression `u * pitch + col` with pitch = RB and col = rb_col(s, units).  What the weight-gradient GEMM
 * reads per k-step -- 64 samples of 128 rows -- is then ONE contiguous 32 KB (16 KB in the bf16 format) instead of 128
 * segments of 256 B each 2 MB from the next (a DRAM row and a page per segment: 3.7 TB/s; blocked: 4.8 TB/s), and what a
 * workgroup of the chain kernels writes per layer is contiguous as well.  `units` is the unit count of the whole matrix
 * (the block stride); sizes are unchanged (rows x pitch dwords, pitch a multiple of 128). */
constexpr int RB = 64;
__host__ __device__ inline long long rb_col(long long s, int units) { return (s >> 6) * ((long long)units * RB) + (s & (RB - 1)); }
constexpr int DEL_SP = 0;                          /* 8 x 256: spatial layer deltas         */
constexpr int DEL_HEADS = DEL_SP + 8 * WIDTH;      /* 144: head rows (HROW_* order), 139 used */
constexpr int DEL_VD = DEL_HEADS + 144;            /* 8 x 256: directional layer deltas     */
constexpr int DEL_RGB = DEL_VD + 8 * WIDTH;        /* 4: rgb layer (3 used)                 */
constexpr int DEL_ROWS = DEL_RGB + 4;              /* 4244 */
/* unit counts (block strides) of the two matrices in the two formats; the bf16 ACT block region ends where the
 * sample-major block begins */
/* (ODD counts: the block stride is then an odd number of 256-B lines and the same unit of successive blocks -- what
 * all workgroups of a chain kernel write at the same moment -- rotates through the memory channels; the matrices are
 * allocated one row larger for it: ACT_ALLOC_ROWS / DEL_ALLOC_ROWS) */
constexpr int ACT_UNITS_F32 = ACT_ROWS_TOTAL + 1, ACT_UNITS_H16 = SMB_ROW0, DEL_UNITS_F32 = DEL_ROWS + 1, DEL_UNITS_H16 = DEL_ROWS / 2 + 1;
constexpr int ACT_ALLOC_ROWS = ACT_ROWS_TOTAL + 1, DEL_ALLOC_ROWS = DEL_ROWS + 1;
static_assert((ACT_UNITS_F32 & ACT_UNITS_H16 & DEL_UNITS_F32 & DEL_UNITS_H16 & 1) == 1, "odd block strides");
constexpr int act_units(bool h16) { return h16 ? ACT_UNITS_H16 : ACT_UNITS_F32; }
constexpr int del_units(bool h16) { return h16 ? DEL_UNITS_H16 : DEL_UNITS_F32; }
/* Split-f16 formats (REFNERF_ACT_F16X2: written by the split-f16 training forward, read by the split-f16 backward and its
 * weight-gradient GEMM; round 4).  Same bytes per ACT element, half the bytes per DELTA element, and no arithmetic between the
 * chain kernels' registers and HBM:
 *   ACT   : the fp32 format's unit grid (ACT_UNITS_F32) with the rows taken in PAIRS -- unit 2j holds the packed HI halves of
 *           rows (2j, 2j + 1), unit 2j + 1 their packed LO halves (x = hi + lo, 22 bits): exactly the dwords of the chain
 *           kernels' packed B fragments, stored as they are; the mask rows keep their raw dwords;
 *   DELTA : pair-rows of ONE IEEE half per element (DEL_ROWS / 2 units: the hi halves of the backward's packed deltas, i.e.
 *           delta * c_s rounded to 11 bits, with c_s the power-of-two factor the chain carries for sample s in that layer),
 *           followed by DSC_ROWS fp32 units holding c_s per (layer id, sample).  The weight-gradient GEMM brings every sample
 *           of a layer to the layer's smallest factor (the largest deltas keep all their bits, the others shift down inside
 *           the half's 40 binades) and divides the tile by it at the end.  Layer ids: spatial 0..7, heads 8, directional
 *           9..16, rgb 17. */
constexpr int DSC_ROWS = 18, DSC0 = DEL_ROWS / 2;
constexpr int DEL_UNITS_F16S = DSC0 + DSC_ROWS + 1;
static_assert((DEL_UNITS_F16S & 1) == 1 && DEL_UNITS_F16S <= DEL_ALLOC_ROWS, "odd block stride inside the fp32-sized allocation");
constexpr int del_layer_id(int d_row) {
  return d_row >= DEL_RGB ? 17 : (d_row >= DEL_VD ? 9 + (d_row - DEL_VD) / WIDTH : (d_row >= DEL_HEADS ? 8 : (d_row - DEL_SP) / WIDTH));
}

/* ---------------- bf16 MFMA operand image ----------------
 * Same 18 ops on v_mfma_f32_32x32x16_bf16 (K = 16 per step), two 32-sample
 * blocks per wave.  Weights travel HBM/L2 -> LDS once per workgroup as uniform
 * "chunks" of 17 KB = [1 KB bias piece][16 fragment pieces of 1 KB], stored in
 * EXECUTION ORDER so the LDS-DMA prefetcher just walks the image.
 * A slice = (op, 32-row output block ob) = 1 or 2 chunks:
 *   REG   : 16 k-steps over the previous layer's packed accumulators
 *   LDS8  : 8 k-steps over encodings staged in LDS (IPE: 6 real + 2 zero steps)
 *   BNLDS : 8 k-steps over the bottleneck (kept in registers) + 8 k-steps over
 *           the dir encodings in LDS (5 real + 3 zero steps)
 * sp0: [LDS8]; sp5: [REG][LDS8]; vd0: [BNLDS]; vd5: [REG][BNLDS]; else [REG].
 * Fragment piece t: lane l holds 8 bf16 = W[32*ob + (l&31)][kmap(t, l>>5, e)],
 * e = 0..7.  Register steps: kmap = 32*(t>>1) + (r&3) + 8*(r>>2) + 4*h with
 * r = 8*(t&1) + e (the accumulator layout of the producing MFMA); LDS steps:
 * k' = 16*t + 8*h + e with IPE k' = canonical feature index and dir k' =
 * [Re(IDE) x36 | n.v | 0 0 0 | Im(IDE) x36 | 0 0 0 0].  Bias piece (first chunk of a slice): fp32
 * [h][16] in accumulator layout. */
constexpr int BF_CHUNK_KB = 17;
constexpr int BF_CHUNK_BYTES = BF_CHUNK_KB * 1024;
constexpr int BF_IPE_REAL_KS = 6;   /* 96 / 16 */
constexpr int BF_DIR_REAL_KS = 5;   /* 80 / 16 */
enum { BF_REG = 0, BF_LDS8 = 1, BF_BNLDS = 2 };

struct BfOp { int nob; int nchunk; int kind[2]; int chunk0; };
struct BfPacked { BfOp op[NUM_OPS]; int chunks_per_pass; };

constexpr BfPacked make_bf_packed() {
  BfPacked P{};
  int c = 0;
  for (int i = 0; i < NUM_OPS; ++i) {
    BfOp o{};
    o.nob = (i == OP_HEADS) ? 5 : (i == OP_RGB ? 1 : 8);
    if (i == 0) { o.nchunk = 1; o.kind[0] = BF_LDS8; }
    else if (i == 5) { o.nchunk = 2; o.kind[0] = BF_REG; o.kind[1] = BF_LDS8; }
    else if (i == 9) { o.nchunk = 1; o.kind[0] = BF_BNLDS; }
    else if (i == 14) { o.nchunk = 2; o.kind[0] = BF_REG; o.kind[1] = BF_BNLDS; }
    else { o.nchunk = 1; o.kind[0] = BF_REG; }
    o.chunk0 = c;
    c += o.nob * o.nchunk;
    P.op[i] = o;
  }
  P.chunks_per_pass = c;
  return P;
}
constexpr BfPacked BFPACKED = make_bf_packed();

/* ---------------- split-f16 operand image (REFNERF_PREC_F16X2) ----------------
 * The parity-grade 16-bit mode: the spatial trunk and the scalar head block (density) take BOTH operands as
 * hi + lo pairs of IEEE halves (x = hi + lo exactly to 22 bits), the directional trunk stays plain f16 (measured on the
 * trained-like weights, scripts/exp_split_precision.py: only the density path needs more than 11 bits).
 * A pass = the spatial section TWICE (two runs of 16 samples per wave), then the directional section once over all 32.
 * The spatial section runs on v_mfma_f32_16x16x32_f16 with THREE partial products (round 4; round 3's N-packed 32x32x16 form
 * -- columns [hi of 16 samples | lo of 16 samples] -- paid W_lo x [x_hi | x_lo] for the W_lo x_hi it needs: 4 products; a
 * K-concatenated 3-product form on the 32x32 shape needs the hi AND lo halves of 32 samples resident, 256 registers).  A
 * wave's 16 samples are the 16 columns: a layer's input is 8 k-steps of 32 as an H (hi halves) and an L (lo halves) fragment set
 * -- 64 registers -- and W_hi H + W_lo H + W_hi L go into ONE accumulator (no cross-lane sums in the epilogue).  A
 * 32-row slice = two 16-row tiles (T0, T1); lane (b = lane / 16, n = lane % 16) of an accumulator holds rows 4 b .. 4 b + 3 of
 * its tile for sample n, and those eight values of a slice are k-group b of the NEXT layer's k-step `slice`:
 *     feature of (k-step s, group b, element i) = 32 s + 16 (i / 4) + 4 b + i % 4.
 * Chunks (same 17 KB: bias piece [T][b][4] fp32 + 16 pieces of 1 KB, piece lane (b, r) = 8 halves W[16 T + r][k-group b]):
 *   SQ_A / SQ_B : k-steps 0..3 / 4..7, pieces [WhT0 WhT1 WlT0 WlT1] x 4, 24 MFMAs of 16.5 cycles
 *   SQ_X        : the 3 IPE k-steps (96 features, k' = 32 s + 8 b + i) from the LDS planes, 12 pieces, 18 MFMAs
 *   SQ_BN       : a bottleneck slice: hi weights only over H and L, pieces [WhT0 WhT1] x 8, 32 MFMAs
 *   SQ_SC       : the scalar head block (rows 128..139: tile T0 only), pieces [WhT0 WlT0] x 8, 24 MFMAs
 * sp0: [SQ_X]; sp1-4, 6, 7: [SQ_A][SQ_B]; sp5: [SQ_A][SQ_B][SQ_X]; heads: 4 x [SQ_BN] + [SQ_SC].  The two runs' bottlenecks
 * merge into the directional trunk's 32-sample fragments with one v_permlane16_swap per dword: slice ob becomes its k-steps
 * 2 ob, 2 ob + 1 with feature (t, h, e) = 32 (t / 2) + 16 (e / 4) + 8 h + 4 (t % 2) + e % 4 (the BNLDS chunks of this image). */
enum { SQ_A = 6, SQ_B = 7, SQ_X = 8, SQ_BN = 9, SQ_SC = 10 };
struct SpPacked { int chunk0[NUM_OPS]; int sp_chunks; int total_chunks; int chunks_per_pass; };
constexpr int sp_slice_chunks(int op, int ob) {
  if (op == 0) return 1;
  if (op == 5) return 3;
  if (op < OP_HEADS) return 2;
  if (op == OP_HEADS) return 1;
  return (op == 14) ? 2 : 1;
}
constexpr SpPacked make_sp_packed() {
  SpPacked P{};
  int c = 0;
  for (int i = 0; i < NUM_OPS; ++i) {
    P.chunk0[i] = c;
    const int nob = (i == OP_HEADS) ? 5 : (i == OP_RGB ? 1 : 8);
    for (int ob = 0; ob < nob; ++ob) c += sp_slice_chunks(i, ob);
    if (i == OP_HEADS) P.sp_chunks = c;
  }
  P.total_chunks = c;
  P.chunks_per_pass = c + P.sp_chunks;
  return P;
}
constexpr SpPacked SPPACKED = make_sp_packed();
static_assert(SPPACKED.sp_chunks == 133 && SPPACKED.total_chunks == 206, "split image layout");

/* head rows inside op 8 (5 blocks of 32): 0..127 bottleneck, then */
constexpr int HROW_DENSITY = 128, HROW_GRAD = 129, HROW_ROUGH = 132, HROW_DIFFUSE = 133, HROW_TINT = 136, HROWS = 139;

}  // namespace rn
