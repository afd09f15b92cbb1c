// Micro-benchmark 2: the two waves of each SIMD run half a chunk out of phase (4-slot ring):
// group A (waves 0-3): [k 0..7 of chunk c] BAR [k 8..15 of chunk c]
// group B (waves 4-7): [k 8..15 of chunk c-1] BAR [k 0..7 of chunk c]
// so that one group's rendezvous / DMA issue / epilogue sits beside the other's MFMA burst.
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
#define AFD 4
#endif
#ifndef EPI
#define EPI 0
#endif
constexpr int CHUNK = 17 * 1024;
constexpr int NCH = 272;
constexpr int FR0 = 1024;

__global__ __launch_bounds__(512) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool grpB = wave >= 4;
  char *WB = smem;
  const char *src = img + wave * 3072 + lane * 16;
  const char *src_end = src + (size_t)136 * CHUNK;
  int left = NCH, nissued = 0;
  auto issue = [&]() {          // next chunk of the stream into slot (nissued & 3)
    if (left > 0) {
      const int slot = (nissued & 3) * CHUNK;
#ifdef DMAHI
      if (wave >= 4) {
        lptr_t dst = (lptr_t)(WB + slot + (wave - 4) * 4096);
        const char *s2 = src - wave * 3072 + (wave - 4) * 4096;
        __builtin_amdgcn_global_load_lds((gptr_t)s2, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)s2, dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)s2, dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)s2, dst, 16, 3072, 0);
        if (wave == 7) __builtin_amdgcn_global_load_lds((gptr_t)s2, dst, 16, 4096, 0);
      }
#else
      if (wave < 6) {
        lptr_t dst = (lptr_t)(WB + slot + wave * 3072);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
        if (wave < 5) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
      }
#endif
      src += CHUNK;
      if (src == src_end) src -= (size_t)136 * CHUNK;
      left -= 1;
    }
    nissued += 1;
  };
  issue(); issue(); issue();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // fragment stream index f = 16*chunk + k
  auto faddr = [&](int f) { return WB + ((f >> 4) & 3) * CHUNK + FR0 + (f & 15) * 1024 + lane * 16; };
  v8bf a[AFD];
  int fnext = 0;                 // next fragment to fetch
#pragma unroll
  for (int d = 0; d < AFD; ++d) { a[d] = *reinterpret_cast<const v8bf *>(faddr(fnext)); ++fnext; }
  v4u bfrag[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bfrag[i] = (v4u){0x3f803f80u + ((lane * 7 + i * 3) & 0x7f), 0x3f803f80u, 0x3f003f80u + i, 0x3f803f80u};
  v16f acc0 = {0};
  float sink = 0.f;
  auto half = [&](int k0) {      // 8 k-steps k0..k0+7 of the current slice
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = k0 + kk;
      v8bf b = __builtin_bit_cast(v8bf, bfrag[k]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % AFD], b, acc0, 0, 0, 0);
      a[kk % AFD] = *reinterpret_cast<const v8bf *>(faddr(fnext)); ++fnext;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto epilogue = [&]() {
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += acc0[r];
#pragma unroll
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 1.0f;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto rendezvous = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue();
  };
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  if (grpB) {
    half(0);                                        // first half of chunk 0, un-synchronised
#pragma unroll 1
    for (int c = 1; c < NCH; ++c) { half(8); epilogue(); rendezvous(); half(0); }
    half(8); epilogue(); rendezvous();
  } else {
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) { half(0); rendezvous(); half(8); epilogue(); }
  }
  long long t1 = __builtin_readcyclecounter();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if (wave == 0 && lane == 0) { cyc[8 + 2 * blockIdx.x] = t1 - t0; cyc[9 + 2 * blockIdx.x] = r1 - r0; }
  if (sink == 12345.678f) out[tid] = sink;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main(int argc, char **argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 2048;
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
  size_t lds = 4 * CHUNK + 70000;
  hipFuncSetAttribute((const void *)mlp_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(512), lds, 0, img, out, cyc);
  hipEventRecord(e0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(mlp_loop, dim3(grid), dim3(512), lds, 0, img, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  { std::vector<long long> pb(2 * (size_t)grid); hipMemcpy(pb.data(), cyc + 8, 16 * (size_t)grid, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0; for (int i = 0; i < grid; ++i) { sc += pb[2 * i]; sr += pb[2 * i + 1]; }
    printf("[all blocks: loop %.0f cycles/chunk, %.3f us/chunk, clock %.2f GHz, loops sum/CU %.3f ms] ", sc / grid / NCH, sr / grid / NCH / 100.0, sc / sr / 10.0, sr / 100.0 / 256 / 1000.0); }
  double flop = (double)grid * 8 * NCH * 16 * 32768.0;
  printf("%s grid %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)  cycles/chunk wave0 %.0f wave4 %.0f  err=%s\n", VARIANT, grid, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)h[0] / NCH, (double)h[4] / NCH, hipGetErrorString(hipGetLastError()));
  return 0;
}
