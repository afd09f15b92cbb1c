// Micro-benchmark of the bf16 level kernel's MLP inner loop (LDS-DMA chunk ring + A fragments
// from LDS + MFMA chain + per-chunk rendezvous), to find what bounds it.  Build variants with -D:
//   NODMA   : no LDS-DMA (fragments re-read from a static ring)
//   NOBAR   : no rendezvous
//   DUAL    : even/odd k-steps into two accumulators
//   AFD=n   : A-fragment ring depth
//   PRIO    : s_setprio 1 for waves 4-7
//   EPI=n   : n dummy VALU ops after each chunk (epilogue stand-in)
//   STAG    : waves 4-7 issue their DMA 4 k-steps later
//   PIECES16: 16-piece (16 KB) chunks, 2 pieces per wave
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
#ifdef PIECES16
constexpr int CHUNK = 16 * 1024;
#else
constexpr int CHUNK = 17 * 1024;
#endif
constexpr int NCH = 272;      // chunks per pass (2 x 136)

__global__ __launch_bounds__(512) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char *WB = smem;
  int cur = 0, nxt = CHUNK, fil = 2 * CHUNK;
#ifdef BAR2
  int fil2 = 3 * CHUNK;
#endif
#if defined(DMAHI)
  const char *src = img + (wave - 4) * 4096 + lane * 16;
#elif defined(DMALO)
  const char *src = img + wave * 4096 + lane * 16;
#elif defined(PIECES16)
  const char *src = img + wave * 2048 + lane * 16;
#else
  const char *src = img + wave * 3072 + lane * 16;
#endif
  const char *src_end = src + (size_t)136 * CHUNK;
  int left = NCH;
#ifdef ROT
  const int rot = (blockIdx.x * 5) % 17;
  const char *srcb = img + lane * 16;
  const char *srcb_end = srcb + (size_t)136 * CHUNK;
#endif
  auto issue = [&](int slot) {
    if (left > 0) {
#ifndef NODMA
#if defined(ROT)
      // piece order rotated per workgroup: at any instant different CUs pull different L2 lines
      if (wave < 6) {
        const int np = (wave < 5) ? 3 : 2;
        for (int q = 0; q < np; ++q) {
          int piece = (wave * 3 + q + rot) % 17;
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + piece * 1024), (lptr_t)(WB + slot + piece * 1024), 16, 0, 0);
        }
      }
      srcb += CHUNK;
      if (srcb == srcb_end) srcb -= (size_t)136 * CHUNK;
#elif defined(DMAHI)
      // all 17 pieces issued by the prioritized waves 4-7 (4,4,4,5): they idle at the rendezvous anyway
      if (wave >= 4) {
        lptr_t dst = (lptr_t)(WB + slot + (wave - 4) * 4096);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 3072, 0);
        if (wave == 7) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 4096, 0);
      }
#elif defined(DMALO)
      if (wave < 4) {
        lptr_t dst = (lptr_t)(WB + slot + wave * 4096);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 3072, 0);
        if (wave == 3) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 4096, 0);
      }
#elif defined(PIECES16)
      lptr_t dst = (lptr_t)(WB + slot + wave * 2048);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
#else
      if (wave < 6) {
        lptr_t dst = (lptr_t)(WB + slot + wave * 3072);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
        if (wave < 5) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
      }
#endif
#endif
      src += CHUNK;
      if (src == src_end) src -= (size_t)136 * CHUNK;
      left -= 1;
    }
  };
#ifdef VLOAD
  // weight stream through VGPRs: wave w owns pieces 2w, 2w+1 (+ piece 16 for wave 0)
  v4u st0 = {0}, st1 = {0}, st2 = {0};
  const char *vsrc = img + wave * 2048 + lane * 16;
  const char *vsrc_end = vsrc + (size_t)136 * CHUNK;
  int vleft = NCH - 2, wslot = fil;
  auto vload = [&]() {
    if (vleft > 0) {
      st0 = *reinterpret_cast<const v4u *>(vsrc);
      st1 = *reinterpret_cast<const v4u *>(vsrc + 1024);
      if (wave == 0) st2 = *reinterpret_cast<const v4u *>(vsrc + 16384);
      vsrc += CHUNK; if (vsrc == vsrc_end) vsrc -= (size_t)136 * CHUNK;
    }
  };
  auto vwrite = [&](int slot) {
    if (vleft > 0) {
      char *d = WB + slot + wave * 2048 + lane * 16;
      *reinterpret_cast<v4u *>(d) = st0;
      *reinterpret_cast<v4u *>(d + 1024) = st1;
      if (wave == 0) *reinterpret_cast<v4u *>(d + 16384) = st2;
      vleft -= 1;
    }
  };
#endif
  issue(cur); issue(nxt);
#ifdef BAR2
  issue(fil);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  constexpr int FR0 = CHUNK - 16 * 1024;     // first fragment piece
  v8bf a[AFD];
#pragma unroll
  for (int d = 0; d < AFD; ++d) a[d] = *reinterpret_cast<const v8bf *>(WB + FR0 + lane * 16 + d * 1024);
  v4u bfrag[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bfrag[i] = (v4u){0x3f803f80u + ((lane * 7 + i * 3) & 0x7f), 0x3f803f80u, 0x3f003f80u + i, 0x3f803f80u};
  v16f acc0 = {0}, acc1 = {0};
  float sink = 0.f;
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const char *pc = WB + cur + FR0 + lane * 16;
    const char *pn = WB + nxt + FR0 + lane * 16;
#ifdef DUAL
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 1.0f; acc1[r] = 0.0f; }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 1.0f;
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v8bf b = __builtin_bit_cast(v8bf, bfrag[k]);
#ifdef DUAL
      if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], b, acc0, 0, 0, 0);
#else
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], b, acc0, 0, 0, 0);
#endif
      a[k % AFD] = (k + AFD < 16) ? *reinterpret_cast<const v8bf *>(pc + (k + AFD) * 1024)
                                  : *reinterpret_cast<const v8bf *>(pn + (k + AFD - 16) * 1024);
#if defined(ENDBAR)
      // waves 4-7 meet the others at the END of their chunk instead of its middle: the two waves of a SIMD run
      // half a chunk out of phase (same barrier count, same 3-slot ring; they refill the slot they just left)
      if (wave < 4 ? k == 7 : k == 15) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(wave < 4 ? fil : cur);
      }
#elif defined(BAR2)
      // 4-slot ring: rendezvous every second chunk, two chunks issued behind it
      if (k == 7 && (c & 1) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(fil);
      }
      if (k == 7 && (c & 1) == 1) issue(fil2);
#else
#ifdef VLOAD
      if (k == 7) {
        // pieces loaded one rendezvous ago go to the slot freed one rendezvous ago, then the next loads start
        if (c > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); vwrite(wslot); }
        __syncthreads();
        wslot = fil;
        vload();
      }
#else
      if (k == 7) {
#ifndef NOBAR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
#ifdef STAG
        if (wave < 4) issue(fil);
#else
        issue(fil);
#endif
      }
#endif
#endif
#ifdef STAG
      if (k == 11 && wave >= 4) issue(fil);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef DUAL
    acc0 = acc0 + acc1;
#endif
    // epilogue stand-in: consume the tile (forces the drain) + EPI dummy VALU ops
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += acc0[r];
#pragma unroll
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e;
    __builtin_amdgcn_sched_barrier(0);
#ifdef BAR2
    { int t = cur; cur = nxt; nxt = fil; fil = fil2; fil2 = t; }
#else
    int t = cur; cur = nxt; nxt = fil; fil = t;
#endif
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
#ifdef BAR2
  size_t lds = 4 * CHUNK;
#else
  size_t lds = 3 * CHUNK;
#endif
#ifdef LDSPAD
  lds += LDSPAD;
#endif
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
