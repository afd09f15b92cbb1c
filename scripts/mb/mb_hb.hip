// Micro-benchmark 4: half-phase with ONE instruction stream, "group B behind" variant on the shipped 3 x 17 KB ring.
//   every wave, per chunk c:  [rdv if group B]  X(c) = k 0..7  [rdv if group A]  Y(c) = k 8..15  epilogue
// Rendezvous n (the n-th of every wave; A's sits in the middle of chunk n-1, B's at the head of chunk n-1) certifies
// chunk n (issued at rendezvous n-1) and lets the prioritised group issue chunk n+1 into the slot of chunk n-2, which
// group A left when it finished X(n-1)'s predecessor and group B when it finished the epilogue of chunk n-2.
// The conditional rendezvous is ONE inline-asm statement with internal branches (the chunk body stays one basic block).
//   -DCHECK : random weight image, result compared with a ring-less evaluation
//   -DNODMA, -DPRIO, -DEPI=n;  -DLOCKSTEP: both groups rendezvous mid-chunk (the shipped schedule)
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
constexpr int CHUNK = 17 * 1024;          // image: [1 KB bias piece | 16 x 1 KB fragment pieces] per chunk
constexpr int HALF = 9 * 1024;            // LDS half-slot stride
constexpr int NIMG = 136;
#ifndef NSLOT
#define NSLOT 8                   // half-slots in the ring: 6 = one rendezvous interval of DMA flight, 8 = two
#endif                 // chunks in the image (the stream wraps)
#ifndef NCH
#define NCH 272
#endif

// Wave-uniform conditional rendezvous as ONE opaque statement with internal branches.  Every participating wave:
// s_waitcnt vmcnt(0) + s_barrier; the issuing waves (np = 4 or 5 pieces) then move their pieces of the next chunk into the
// free slot by LDS-DMA (buffer_load ... lds; stream position = ONE SGPR) and advance it by one chunk, with wrap.
__device__ __forceinline__ void rendezvous(int go, unsigned np, unsigned d, unsigned &soff, v4u rsrc, unsigned vlane16) {
  unsigned keep, tmp;
  go = __builtin_amdgcn_readfirstlane(go);
  d = __builtin_amdgcn_readfirstlane(d);
  asm volatile(
      "s_cmp_eq_u32 %[go], 0\n\t"
      "s_cbranch_scc1 .Lrdv%=\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_barrier\n\t"
      "s_cmp_eq_u32 %[np], 0\n\t"
      "s_cbranch_scc1 .Lrdv%=\n\t"
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[d]\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %[vl], %[rsrc], %[soff] offen lds\n\t"
      "buffer_load_dwordx4 %[vl], %[rsrc], %[soff] offen offset:1024 lds\n\t"
      "buffer_load_dwordx4 %[vl], %[rsrc], %[soff] offen offset:2048 lds\n\t"
      "buffer_load_dwordx4 %[vl], %[rsrc], %[soff] offen offset:3072 lds\n\t"
      "s_cmp_lt_u32 %[np], 5\n\t"
      "s_cbranch_scc1 .Lrdvm%=\n\t"
      "s_add_u32 m0, %[d], 0x1000\n\t"
      "s_add_u32 %[tmp], %[soff], 0x1000\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %[vl], %[rsrc], %[tmp] offen lds\n\t"
      ".Lrdvm%=:\n\t"
      "s_mov_b32 m0, %[keep]\n\t"
      "s_add_u32 %[soff], %[soff], %[chunk]\n\t"
      "s_cmp_ge_u32 %[soff], %[simg]\n\t"
      "s_cselect_b32 %[tmp], %[simg], 0\n\t"
      "s_sub_u32 %[soff], %[soff], %[tmp]\n\t"
      ".Lrdv%=:\n\t"
      : [keep] "=&s"(keep), [tmp] "=&s"(tmp), [soff] "+s"(soff)
      : [go] "s"(go), [np] "s"(np), [d] "s"(d), [rsrc] "s"(rsrc), [vl] "v"(vlane16), [chunk] "n"(CHUNK), [simg] "n"(NIMG * CHUNK)
      : "memory", "scc");
}

__global__ __launch_bounds__(512) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grpB = wave >= 4 ? 1 : 0;
  char *WB = smem;
  const unsigned wb_lds = (unsigned)(size_t)(lptr_t)WB;
  const unsigned vlane16 = lane * 16;
  const unsigned long long ib = (unsigned long long)img;
  v4u rsrc = {(unsigned)ib, (unsigned)(ib >> 32) & 0xffffu, (unsigned)((NIMG + 4) * CHUNK), 0x00020000u};
#pragma unroll
  for (int i = 0; i < 4; ++i) rsrc[i] = __builtin_amdgcn_readfirstlane(rsrc[i]);
  // prologue: chunks 0 and 1
  for (int q = wave; q < 2 * 17; q += 8) {
    const int c = q / 17, r = q % 17;
    __builtin_amdgcn_global_load_lds((gptr_t)(img + (size_t)c * CHUNK + r * 1024 + lane * 16), (lptr_t)(WB + c * CHUNK + r * 1024), 16, 0, 0);
  }
  // issuing waves 4-7: pieces 4(w-4) .. (+4, wave 7: +5) of the chunk; the first chunk issued is chunk 2 (at rendezvous 1)
  const unsigned p0 = __builtin_amdgcn_readfirstlane(wave >= 4 ? (wave - 4) * 4 : 0);
  unsigned soff = __builtin_amdgcn_readfirstlane(2 * CHUNK + p0 * 1024);
#ifdef NODMA
  const unsigned np_w = 0;
#else
  const unsigned np_w = __builtin_amdgcn_readfirstlane(wave < 4 ? 0 : (wave == 7 ? 5 : 4));
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  v8bf a[AFD];
#pragma unroll
  for (int d2 = 0; d2 < AFD; ++d2) a[d2] = *reinterpret_cast<const v8bf *>(WB + 1024 + d2 * 1024 + lane * 16);
  v4u bfrag[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bfrag[i] = (v4u){0x3f803f80u + ((lane * 7 + i * 3) & 0x7f), 0x3f803f80u, 0x3f003f80u + i, 0x3f803f80u};
  v16f acc0;
  float sink = 0.f;
  int cur = 0, nxt = CHUNK, fil = 2 * CHUNK;
#ifdef LOCKSTEP
  const int goA = 1, goB = 0;
#else
  const int goA = __builtin_amdgcn_readfirstlane(1 - grpB), goB = __builtin_amdgcn_readfirstlane(grpB);
#endif
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const char *pc = WB + cur + lane * 16;
    const char *pnx = WB + nxt + lane * 16;
    const unsigned dfil = wb_lds + fil + p0 * 1024;
    rendezvous(goB, np_w, dfil, soff, rsrc, vlane16);                 // group B: at the head of the chunk
    {
      const float b0 = *reinterpret_cast<const float *>(pc - lane * 16 + (lane >> 5) * 64);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = b0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, bfrag[k]), acc0, 0, 0, 0);
      const int kn = k + AFD;
      a[k % AFD] = (kn < 16) ? *reinterpret_cast<const v8bf *>(pc + 1024 + kn * 1024) : *reinterpret_cast<const v8bf *>(pnx + 1024 + (kn - 16) * 1024);
      if (k == 7) rendezvous(goA, np_w, dfil, soff, rsrc, vlane16);   // group A: between X and Y
      __builtin_amdgcn_sched_barrier(0);
    }
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += acc0[r];
#pragma unroll
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e * (1.0f / 65536.0f);
    __builtin_amdgcn_sched_barrier(0);
    const int t = cur; cur = nxt; nxt = fil; fil = t;
  }
  long long t1 = __builtin_readcyclecounter();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if (wave == 0 && lane == 0) { cyc[8 + 2 * blockIdx.x] = t1 - t0; cyc[9 + 2 * blockIdx.x] = r1 - r0; }
  out[blockIdx.x * 512 + tid] = sink;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

#ifdef CHECK
// the same arithmetic without the ring: every wave reads its fragments straight from the image
__global__ __launch_bounds__(512) void mlp_ref(const char *img, float *out) {
  const int tid = threadIdx.x, lane = tid & 63;
  v4u bfrag[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bfrag[i] = (v4u){0x3f803f80u + ((lane * 7 + i * 3) & 0x7f), 0x3f803f80u, 0x3f003f80u + i, 0x3f803f80u};
  float sink = 0.f;
  for (int c = 0; c < NCH; ++c) {
    const char *ch = img + (size_t)(c % NIMG) * CHUNK;
    v16f acc0;
    const float b0 = *reinterpret_cast<const float *>(ch + (lane >> 5) * 64);
    for (int r = 0; r < 16; ++r) acc0[r] = b0;
    for (int k = 0; k < 16; ++k)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const v8bf *>(ch + 1024 + k * 1024 + lane * 16),
                                                     __builtin_bit_cast(v8bf, bfrag[k]), acc0, 0, 0, 0);
    float e = 0.f;
    for (int r = 0; r < 16; ++r) e += acc0[r];
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e * (1.0f / 65536.0f);
  }
  out[blockIdx.x * 512 + tid] = sink;
}
#endif

int main(int argc, char **argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 2048;
  char *img; float *out, *out2; long long *cyc;
  const size_t img_bytes = (size_t)(NIMG + 4) * CHUNK;
  hipMalloc(&img, img_bytes);
  std::vector<unsigned short> h(img_bytes / 2);
  unsigned s = 12345;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = 0x3c00 + ((s >> 16) & 0x1ff); }   // bf16 values in [2^-7, 2^-5)
  for (size_t c = 0; c < NIMG + 4; ++c)                                     // bias piece: small fp32 values
    for (int i = 0; i < 256; ++i) reinterpret_cast<float *>(h.data() + c * CHUNK / 2)[i] = 0.001f * (float)((c * 31 + i) % 17);
  for (size_t c = NIMG; c < NIMG + 4; ++c)                                  // the stream runs past the wrap point: chunks NIMG.. repeat 0..
    for (size_t i = 0; i < CHUNK / 2; ++i) h[c * (CHUNK / 2) + i] = h[(c - NIMG) * (CHUNK / 2) + i];
  hipMemcpy(img, h.data(), img_bytes, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)grid * 512 * 4); hipMalloc(&out2, (size_t)grid * 512 * 4); hipMalloc(&cyc, 64 + 16 * (size_t)grid);
  size_t lds = 3 * CHUNK;
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
  long long hc[8]; hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
  { std::vector<long long> pb(2 * (size_t)grid); hipMemcpy(pb.data(), cyc + 8, 16 * (size_t)grid, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0; for (int i = 0; i < grid; ++i) { sc += pb[2 * i]; sr += pb[2 * i + 1]; }
    printf("[all blocks: loop %.0f cycles/chunk, %.3f us/chunk, clock %.2f GHz, loops sum/CU %.3f ms] ", sc / grid / NCH, sr / grid / NCH / 100.0, sc / sr / 10.0, sr / 100.0 / 256 / 1000.0); }
  double flop = (double)grid * 8 * NCH * 16 * 32768.0;
  printf("%s grid %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)  cycles/chunk wave0 %.0f wave4 %.0f  err=%s", VARIANT, grid, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 25.0, (double)hc[0] / NCH, (double)hc[4] / NCH, hipGetErrorString(hipGetLastError()));
#ifdef CHECK
  hipLaunchKernelGGL(mlp_ref, dim3(grid), dim3(512), 0, 0, img, out2);
  std::vector<float> a((size_t)grid * 512), b((size_t)grid * 512);
  hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]);
  printf("  check: %zu / %zu differ (sample %g vs %g)", bad, a.size(), a[1], b[1]);
#endif
  printf("\n");
  return 0;
}
