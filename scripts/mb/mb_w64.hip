// Micro-benchmark of a 64-samples-per-wave MLP inner loop: 4 waves (one per SIMD), every A fragment read from LDS
// feeds TWO MFMAs (two 32-sample B tiles) -- half the LDS fragment traffic of the 8 x 32 layout of mb_mlp.hip.
// A "layer" = 8 chunks (slices of 32 outputs x 256 k); the packed outputs of a layer are the B fragments of the next.
//   -DMODE=0 : outputs discarded (structure only: LDS-DMA ring + fragment reads + dual-accumulator MFMA chain)
//   -DMODE=1 : in / out ping-pong as plain arrays (the compiler places them: 256+ registers)
//   -DMODE=2 : outputs parked in AGPRs (v_accvgpr_write), moved back to VGPRs at the layer boundary
//   -DDBUF   : double-buffered accumulators: slice s is packed while slice s+1 runs on the matrix pipe
//   -DNODMA / -DNOBAR as in mb_mlp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#ifndef AFD
#define AFD 4
#endif
#ifndef MODE
#define MODE 0
#endif
constexpr int CHUNK = 17 * 1024;
constexpr int NLAYER = 34;    // 34 layers x 8 chunks = 272 chunks (same work per sample as mb_mlp.hip)

__device__ __forceinline__ unsigned pack_relu(float lo, float hi) {
  v2bf r = __builtin_convertvector((v2f){lo, hi}, v2bf);
  v2s s = __builtin_bit_cast(v2s, r);
  s = __builtin_elementwise_max(s, (v2s){0, 0});
  return __builtin_bit_cast(unsigned, s);
}
__device__ __forceinline__ unsigned to_acc(unsigned v) { unsigned a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ unsigned from_acc(unsigned a) { unsigned v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

__global__ __launch_bounds__(256) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char *WB = smem;
  int cur = 0, nxt = CHUNK, fil = 2 * CHUNK;
  const char *src = img + wave * 4096 + lane * 16;
  const char *src_end = src + (size_t)136 * CHUNK;
  int left = NLAYER * 8;
  auto issue = [&](int slot) {
    if (left > 0) {
#ifndef NODMA
      lptr_t dst = (lptr_t)(WB + slot + wave * 4096);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 3072, 0);
      if (wave == 3) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 4096, 0);
#endif
      src += CHUNK;
      if (src == src_end) src -= (size_t)136 * CHUNK;
      left -= 1;
    }
  };
  issue(cur); issue(nxt);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  constexpr int FR0 = CHUNK - 16 * 1024;
  v8bf a[AFD];
#pragma unroll
  for (int d = 0; d < AFD; ++d) a[d] = *reinterpret_cast<const v8bf *>(WB + FR0 + lane * 16 + d * 1024);
  v8bf dead[AFD];
#pragma unroll
  for (int d = 0; d < AFD; ++d) dead[d] = a[d];
  v4u in[32];          // [k-step][tile]
#pragma unroll
  for (int i = 0; i < 32; ++i) in[i] = (v4u){0x3f803f80u + lane + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#if MODE == 1
  v4u ob[32];
#elif MODE == 2
  unsigned park[128];
#endif
  float sink = 0.f;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int layer = 0; layer < NLAYER; ++layer) {
#ifdef DBUF
    v16f pa0, pa1;     // accumulators of the previous slice, packed while this one runs
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const char *pc = WB + cur + FR0 + lane * 16;
      const char *pn = WB + nxt + FR0 + lane * 16;
      v16f acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 1.0f; acc1[r] = 1.0f; }
#if defined(NOLDS) || defined(DEADLDS)
#pragma unroll
      for (int d = 0; d < AFD; ++d) asm volatile("" : "+v"(a[d]));     // opaque per chunk: nothing is loop-invariant
#endif
      __builtin_amdgcn_sched_barrier(0);
#if defined(GROUP) && defined(HOIST)
      // groups of GROUP k-steps; the GROUP fragment reads of the NEXT group are issued first, then each accumulator runs
      // GROUP dependent MFMAs strictly back-to-back (nothing between two MFMAs on the same accumulator)
      static_assert(AFD == 2 * GROUP, "ring = two groups");
#pragma unroll
      for (int g = 0; g < 16 / GROUP; ++g) {
#pragma unroll
        for (int kk = 0; kk < GROUP; ++kk) {
          const int k = g * GROUP + kk + GROUP;          // fragment of the next group -> the half of the ring group g-1 used
          a[k % AFD] = (k < 16) ? *reinterpret_cast<const v8bf *>(pc + k * 1024) : *reinterpret_cast<const v8bf *>(pn + (k - 16) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int kk = 0; kk < GROUP; ++kk) {
            const int k = g * GROUP + kk;
            if (t == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k]), acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc1, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g == (8 / GROUP) - 1) {
#ifndef NOBAR
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
#endif
          issue(fil);
        }
      }
#elif defined(QUAD)
      // four accumulators: two tiles x even / odd k-steps, round-robin: an accumulator is reused every fourth MFMA
      v16f acc2, acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc2[r] = 0.0f; acc3[r] = 0.0f; }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k & 1) {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k]), acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc3, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc1, 0, 0, 0);
        }
        a[k % AFD] = (k + AFD < 16) ? *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024)
                                    : *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
        if (k == 7) {
#ifndef NOBAR
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
#endif
          issue(fil);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc0 = acc0 + acc2; acc1 = acc1 + acc3;
#elif defined(GROUP)
      // k-steps in groups of GROUP: each accumulator runs GROUP dependent MFMAs back-to-back, the A fragments of a group
      // are shared by both tiles (ring of 2 x GROUP fragments)
      static_assert(AFD == 2 * GROUP, "ring = two groups");
#pragma unroll
      for (int g = 0; g < 16 / GROUP; ++g) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int kk = 0; kk < GROUP; ++kk) {
            const int k = g * GROUP + kk;
            if (t == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k]), acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc1, 0, 0, 0);
            if (t == 1) a[k % AFD] = (k + AFD < 16) ? *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024)
                                                    : *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (g == (8 / GROUP) - 1) {
#ifndef NOBAR
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
#endif
          issue(fil);
        }
      }
#else
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k]), acc0, 0, 0, 0);
#ifdef ONEACC
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc0, 0, 0, 0);
#else
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, in[2 * k + 1]), acc1, 0, 0, 0);
#endif
#if defined(NOLDS)
#elif defined(DEADLDS)
        dead[k % AFD] = (k + AFD < 16) ? *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024)
                                       : *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
        asm volatile("" :: "v"(dead[(k + 1) % AFD]));
#else
        a[k % AFD] = (k + AFD < 16) ? *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024)
                                    : *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
#endif
#ifdef DBUF
        // previous slice's pack, one register pair per k-step, in the shadow of this slice's MFMAs
        if (s > 0) {
          const int e = k & 7;
          const v16f &p = (k < 8) ? pa0 : pa1;
          unsigned w = pack_relu(p[2 * e], p[2 * e + 1]);
#if MODE == 0
          sink += __builtin_bit_cast(float, w);
#elif MODE == 1
          ob[4 * (s - 1) + (k >> 2)][k & 3] = w;
#else
          park[16 * (s - 1) + k] = to_acc(w);
#endif
        }
#endif
        if (k == 7) {
#ifndef NOBAR
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
#endif
          issue(fil);
        }
#ifndef DBUF
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#endif
#ifdef DBUF
      pa0 = acc0; pa1 = acc1;
#else
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned w0 = pack_relu(acc0[2 * e], acc0[2 * e + 1]), w1 = pack_relu(acc1[2 * e], acc1[2 * e + 1]);
#if MODE == 0
        sink += __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
#elif MODE == 1
        ob[4 * s + (e >> 2)][e & 3] = w0; ob[4 * s + 2 + (e >> 2)][e & 3] = w1;
#else
        park[16 * s + e] = to_acc(w0); park[16 * s + 8 + e] = to_acc(w1);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
      int t = cur; cur = nxt; nxt = fil; fil = t;
    }
#ifdef DBUF
    // last slice of the layer
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned w0 = pack_relu(pa0[2 * e], pa0[2 * e + 1]), w1 = pack_relu(pa1[2 * e], pa1[2 * e + 1]);
#if MODE == 0
      sink += __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
#elif MODE == 1
      ob[28 + (e >> 2)][e & 3] = w0; ob[30 + (e >> 2)][e & 3] = w1;
#else
      park[112 + e] = to_acc(w0); park[120 + e] = to_acc(w1);
#endif
    }
#endif
#if MODE == 1
#pragma unroll
    for (int i = 0; i < 32; ++i) in[i] = ob[i] | (v4u){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#elif MODE == 2
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) in[i][j] = from_acc(park[4 * i + j]) | 0x3f803f80u;
#endif
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned fold = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) fold ^= in[i][0] ^ in[i][1] ^ in[i][2] ^ in[i][3];
  if (sink == 12345.678f || fold == 0x12345678u) out[tid] = sink;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main(int argc, char **argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 2048;
  char *img; float *out; long long *cyc;
  hipMalloc(&img, (size_t)140 * 17 * 1024); hipMemset(img, 0x3c, (size_t)140 * 17 * 1024);
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  size_t lds = 3 * CHUNK;
#ifdef LDSPAD
  lds += LDSPAD;
#endif
  hipFuncSetAttribute((const void *)mlp_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(256), lds, 0, img, out, cyc);
  hipEventRecord(e0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(256), lds, 0, img, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  double flop = (double)grid * 4 * NLAYER * 8 * 32 * 32768.0;
  printf("%s grid %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)  cycles/chunk wave0 %.0f wave3 %.0f  err=%s\n", VARIANT, grid, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)h[0] / (NLAYER * 8), (double)h[3] / (NLAYER * 8), hipGetErrorString(hipGetLastError()));
  return 0;
}
