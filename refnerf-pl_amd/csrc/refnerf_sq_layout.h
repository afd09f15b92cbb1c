/*
 * refnerf_sq_layout.h -- layouts of the round-5 training path of the parity-grade 16-bit mode (REFNERF_PREC_F16X2 on the
 * built-in IPE basis): the training forward and the backward run on the EVAL kernel's skeleton (8 waves, two per SIMD, weights
 * through the LDS-DMA chunk ring of refnerf_level_bf16.h, 0 B of scratch) instead of the one-wave-per-SIMD fp32 skeleton.
 *
 * Arithmetic (scripts/exp_train_sq_precision.py: the reference's autograd with these roundings emulated, three trained sets):
 *   forward   spatial trunk + ALL head rows : W and x as hi + lo halves, three products (the eval kernel's; the bottleneck now
 *                                             takes its W_lo product too: LLFF gradient 1.5e-4 -> 6.5e-5)
 *             directional trunk + rgb       : W as hi + lo, x as ONE half: two products [W_hi | W_lo] x (what matters is the
 *                                             coherent rounding of W, not the per-sample rounding of x: W11 x X11 3.8e-4,
 *                                             W22 x X11 1.7e-4 = the all-22-bit figure)
 *   VJP       density-gradient normals      : three products (the normals are an output)
 *   backward  every transposed layer        : W^T as hi + lo, delta as ONE half after its per-sample power-of-two factor: two
 *                                             products (gradient rel-L2 unchanged within the sets' spread)
 *   dW        ACT spatial hi + lo (pair units), ACT directional ONE half (it IS what the forward multiplied), DELTA one half.
 *
 * (1) the weight image (REFNERF_IMAGE_F16X2_TRAIN): 17 KB chunks in execution order, forward stream then backward stream;
 * (2) the ACT format REFNERF_ACT_SQ and the DELTA format (factor rows doubled);
 * (3) the job table of the weight-gradient GEMM on them.
 */
#pragma once
#include "refnerf_layout.h"

namespace rn {

/* ---------------- (1) weight image ----------------
 * FORWARD stream of a pass = [run section] x 2 (16 samples per wave each) + [directional section] (32 samples per wave):
 *   run section  : spatial ops 0..7 exactly as the eval image (SQ_X / SQ_A / SQ_B chunks, 128 chunks)
 *                  heads: 4 bottleneck slices as [SQ_A SQ_B] (hi + lo weights) + the scalar block [SQ_SC]            9
 *                  VJP  : transposed spatial layers 7..1 (8 slices x [SQ_A SQ_B]); behind layer 5 the 96 IPE rows of layer 5
 *                         (3 slices), behind layer 1 those of layer 0 (3 slices)                                  124
 *   directional  : ops 9..16 + rgb as plain f16 chunks of refnerf_layout.h, every chunk twice: [hi][lo]              146
 * BACKWARD stream of a pass (32 samples per wave, two products): plain chunks [hi][lo] of the transposed ops
 *   directional layers 7..1 (8 slices x [REG REG])                                                                112
 *   the 204 dir-input rows: 7 slices x [layer 5's part hi lo | layer 0's part hi lo]                                28
 *   heads^T: 8 slices x [BNLDS hi lo] (K = 128 bottleneck rows from registers + the 11 scalar rows from LDS)         16
 *   spatial layers 7..1                                                                                           112
 * then a constants block (fp32): W_density[256], the column-sum bounds G of every transposed op, W_rgb[3][256]. */
constexpr int TR_SP_TRUNK = 128, TR_HEADS = 9, TR_VJP = 124;
constexpr int TR_RUN = TR_SP_TRUNK + TR_HEADS + TR_VJP;       /* 261 */
constexpr int TR_DIR = 146;
constexpr int TR_FWD = TR_RUN + TR_DIR;                       /* 407 chunks stored  */
constexpr int TR_FWD_PASS = 2 * TR_RUN + TR_DIR;              /* 668 chunks streamed per pass */
constexpr int TR_BWD_DIR = 112, TR_BWD_DIN = 28, TR_BWD_HEADS = 16, TR_BWD_SP = 112;
constexpr int TR_BWD = TR_BWD_DIR + TR_BWD_DIN + TR_BWD_HEADS + TR_BWD_SP;   /* 268 */
constexpr int TR_BWD0 = TR_FWD;
constexpr int TR_CHUNKS = TR_FWD + TR_BWD;                    /* 675 */
static_assert(SPPACKED.chunk0[OP_HEADS] == TR_SP_TRUNK, "the spatial trunk of the eval image");
/* constants block behind the chunks (+ 2 chunks of pad: the DMA never runs past the end, the fragment ring reads one piece ahead) */
constexpr size_t TR_CONST_OFF = (size_t)(TR_CHUNKS + 2) * BF_CHUNK_BYTES;
constexpr int TRC_WD = 0;                  /* raw_density.weight[256]                                   */
constexpr int TRC_G = 256;                 /* G of the transposed ops, indexed by TRG_*                 */
constexpr int TRC_WRGB = 320;              /* rgb_layer.weight[3][256]                                  */
constexpr int TRC_FLOATS = TRC_WRGB + 3 * WIDTH;
constexpr size_t TR_IMAGE_BYTES = TR_CONST_OFF + (size_t)TRC_FLOATS * 4;
/* G[op] = max over the op's output rows (= the layer's input features f) of sum_o |W[o][f]|: |W^T delta|_inf <= G |delta|_inf,
 * so a sample's deltas can be rescaled BEFORE the contraction without ever leaving the range of an IEEE half */
enum { TRG_SP = 0 /* + layer 1..7 */, TRG_SP5_IPE = 8, TRG_SP0 = 9, TRG_VD = 10 /* + layer 1..7 */, TRG_VD5_DIN = 18, TRG_VD0 = 19,
       TRG_HEADS = 20, TRG_RGB = 21, TRG_WD = 22, TRG_N = 23 };

/* ---------------- (2) ACT / DELTA ----------------
 * REFNERF_ACT_SQ: blocked units as every ACT format ([64-sample block][unit][64 samples], one dword per unit and sample):
 *   AQ_IPE  96 units : pair units of the 96 IPE rows (unit 2j = packed hi halves of rows 2j, 2j + 1, unit 2j + 1 their lo halves)
 *   AQ_SP 2048 units : pair units of x0..x7 (the inputs of spatial layers 1..7 and of the heads)
 *   AQ_DIN 102 units : ONE half per element, rows in pairs: [bottleneck 128 | IDE 72 | n.v | 0 0 0]
 *   AQ_VD 1024 units : the same for v0..v7
 *   AQ_MASK 128 units: ReLU sign patterns, lane-local words of the kernels (8 per layer):
 *                      spatial layer l (16-sample tiles: lane (b, n)): unit 8 l + 2 b + j, bit 8 (s % 4) + i of word j = s / 4
 *                        <-> feature 32 s + 16 (i / 4) + 4 b + i % 4;
 *                      directional layer l (32-sample tiles: lane (h, n)): unit 64 + 8 l + 4 h + q, bit 16 (ob % 2) + r of word
 *                        q = ob / 2 <-> feature 32 ob + (r & 3) + 8 (r >> 2) + 4 h
 *   AQ_RAW  16 units : fp32: the 11 raw scalar head rows (HROW_DENSITY..), raw rgb[3]                         13.7 KB / sample */
constexpr int AQ_IPE = 0, AQ_SP = AQ_IPE + IPE_DIM, AQ_DIN = AQ_SP + 8 * WIDTH, AQ_VD = AQ_DIN + DIR_PAD / 2,
              AQ_MASK = AQ_VD + 8 * WIDTH / 2, AQ_RAW = AQ_MASK + 128, AQ_RAW_RGB = AQ_RAW + 11, AQ_UNITS = AQ_RAW + 16 + 1;
static_assert((AQ_UNITS & 1) == 1 && AQ_UNITS <= ACT_ALLOC_ROWS, "odd block stride inside the allocation of the fp32 format");
/* DELTA: the pair-rows of refnerf_layout.h (one half per element, DEL_ROWS / 2 units), then per (layer id, sample) TWO fp32
 * factor units: c (what the stored halves were multiplied by) and kappa = the power of two that would bring the sample's
 * largest |delta| of that layer to [2^14, 2^15) -- the weight-gradient GEMM brings every sample to the layer's smallest kappa */
constexpr int DQ_C = DEL_ROWS / 2, DQ_K = DQ_C + DSC_ROWS, DQ_UNITS = DQ_K + DSC_ROWS + 1;
static_assert((DQ_UNITS & 1) == 1 && DQ_UNITS <= DEL_ALLOC_ROWS, "odd block stride inside the fp32-sized allocation");

}  // namespace rn
