// Micro-benchmark: the 16-bit MLP inner loop as TWO independent 4-wave workgroups per CU (one wave per SIMD each) instead of
// one 8-wave workgroup: the two waves of a SIMD then belong to different workgroups, meet different barriers and drift out
// of phase by themselves (what mb_half.hip / mb_hp.hip arrange by hand inside one workgroup).  Each workgroup streams the
// whole weight image for its 128 samples (2x the L2 -> LDS traffic per CU).  Variants (-D):
//   RING3 : 3-slot ring, rendezvous in the middle of the chunk (as shipped; 51 KB)
//   (default) 2-slot ring, rendezvous at k = 16 - AFD - 1: all fragment reads of the chunk have been issued by then
//            (lgkmcnt(0) + vmcnt(0) + barrier), the slot just left is refilled with the chunk after next (34 KB)
//   EPI=n, RANDIMG, LDSPAD=n as in mb_mlp.hip;  NODMA, NOBAR
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#ifndef AFD
#define AFD 2
#endif
#ifndef EPI
#define EPI 0
#endif
constexpr int CHUNK = 17 * 1024;
constexpr int NCH = 272;
#ifdef RING3
constexpr int SLOTS = 3;
constexpr int KRDV = 7;
#else
constexpr int SLOTS = 2;
constexpr int KRDV = 16 - AFD - 1;
#endif

__global__ __launch_bounds__(256) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char *WB = smem;
  int cur = 0, nxt = CHUNK, fil = (SLOTS == 3) ? 2 * CHUNK : 0;
  const char *src = img + wave * 4096 + lane * 16;
  const char *src_end = src + (size_t)136 * CHUNK;
  int left = NCH;
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
  v4u bfrag[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bfrag[i] = (v4u){0x3f803f80u + ((lane * 7 + i * 3) & 0x7f), 0x3f803f80u, 0x3f003f80u + i, 0x3f803f80u};
  v16f acc0 = {0};
  float sink = 0.f;
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const char *pc = WB + cur + FR0 + lane * 16;
    const char *pn = WB + nxt + FR0 + lane * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 1.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v8bf b = __builtin_bit_cast(v8bf, bfrag[k]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], b, acc0, 0, 0, 0);
      if (k + AFD < 16) a[k % AFD] = *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024);
      else a[k % AFD] = *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
      if (k == KRDV) {
#ifndef NOBAR
#ifdef RING3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* every read of this chunk has landed */
#endif
        __builtin_amdgcn_s_barrier();
#endif
        issue(SLOTS == 3 ? fil : cur);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += acc0[r];
#pragma unroll
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e;
    __builtin_amdgcn_sched_barrier(0);
    if (SLOTS == 3) { int t = cur; cur = nxt; nxt = fil; fil = t; }
    else { int t = cur; cur = nxt; nxt = t; }
  }
  long long t1 = __builtin_readcyclecounter();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if (wave == 0 && lane == 0) { cyc[8 + 2 * blockIdx.x] = t1 - t0; cyc[9 + 2 * blockIdx.x] = r1 - r0; }
  if (sink == 12345.678f) out[tid] = sink;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main(int argc, char **argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 4096;
  char *img; float *out; long long *cyc;
  hipMalloc(&img, (size_t)140 * 17 * 1024);
#ifdef RANDIMG
  { std::vector<unsigned short> hh((size_t)140 * 17 * 512); unsigned sd = 12345;
    for (auto &v : hh) { sd = sd * 1664525u + 1013904223u; v = 0x3c00 + ((sd >> 16) & 0x1ff); }
    hipMemcpy(img, hh.data(), hh.size() * 2, hipMemcpyHostToDevice); }
#else
  hipMemset(img, 0x3c, (size_t)140 * 17 * 1024);
#endif
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64 + 16 * (size_t)grid);
  size_t lds = SLOTS * CHUNK;
#ifdef LDSPAD
  lds += LDSPAD;
#endif
  hipFuncSetAttribute((const void *)mlp_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mlp_loop, 256, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(256), lds, 0, img, out, cyc);
  hipEventRecord(e0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(256), lds, 0, img, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  { std::vector<long long> pb(2 * (size_t)grid); hipMemcpy(pb.data(), cyc + 8, 16 * (size_t)grid, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0; for (int i = 0; i < grid; ++i) { sc += pb[2 * i]; sr += pb[2 * i + 1]; }
    printf("[occ %d; all blocks: loop %.0f cycles/chunk, %.3f us/chunk, clock %.2f GHz] ", occ, sc / grid / NCH, sr / grid / NCH / 100.0, sc / sr / 10.0); }
  double flop = (double)grid * 4 * NCH * 16 * 32768.0;
  printf("%s grid %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)  cycles/chunk wave0 %.0f wave3 %.0f  err=%s\n", VARIANT, grid, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)h[0] / NCH, (double)h[3] / NCH, hipGetErrorString(hipGetLastError()));
  return 0;
}
