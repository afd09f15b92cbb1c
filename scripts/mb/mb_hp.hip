// Micro-benchmark 3: the half-phase schedule with ONE instruction stream for all eight waves.
//   loop over chunks c:  [rdv if group B] X(c) = k 0..7  [rdv if group A]  Y(c) = k 8..15  epilogue
// Group A (waves 0-3) meets the barrier between X and Y, group B (waves 4-7) between the epilogue and the next X, so
// that on every SIMD one wave's rendezvous / DMA issue / epilogue sits beside the other wave's MFMA burst.  The
// conditional rendezvous is ONE inline-asm statement with an internal branch (the chunk body stays one basic block).
// Ring: 6 half-slots of 9 KB (X halves = [bias | frags 0-7] in even slots, Y halves = [frags 8-15] in odd slots): a
// rendezvous issues Y(n+1) and X(n+2), whose slots (those of Y(n-2), X(n-1)) every wave has left.
//   -DCHECK : the weight image is random and the result is compared with a straightforward evaluation (no ring)
//   -DNODMA, -DPRIO, -DEPI=n as in mb_mlp.hip;  -DLOCKSTEP: both groups rendezvous at the same place (baseline)
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

// Wave-uniform conditional rendezvous as ONE opaque statement with internal branches (the caller's chunk body stays one
// basic block).  Every wave: s_waitcnt vmcnt(0) + s_barrier.  The waves with np > 0 (group B: the prioritised half) then
// issue their np pieces of the next (Y, X) pair by LDS-DMA (buffer_load ... lds: the stream position is ONE SGPR, soff) and
// advance their stream state (LDS destination d, image offset soff) by one chunk, with wrap.
__device__ __forceinline__ void rendezvous(int go, unsigned np, unsigned &d, unsigned &soff, unsigned d_end, v4u rsrc, unsigned vlane16) {
  unsigned keep, tmp;
  go = __builtin_amdgcn_readfirstlane(go);
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
      "s_add_u32 %[d], %[d], %[two]\n\t"
      "s_cmp_ge_u32 %[d], %[dend]\n\t"
      "s_cselect_b32 %[tmp], %[ring], 0\n\t"
      "s_sub_u32 %[d], %[d], %[tmp]\n\t"
      "s_add_u32 %[soff], %[soff], %[chunk]\n\t"
      "s_cmp_ge_u32 %[soff], %[simg]\n\t"
      "s_cselect_b32 %[tmp], %[simg], 0\n\t"
      "s_sub_u32 %[soff], %[soff], %[tmp]\n\t"
      ".Lrdv%=:\n\t"
      : [keep] "=&s"(keep), [tmp] "=&s"(tmp), [d] "+s"(d), [soff] "+s"(soff)
      : [go] "s"(go), [np] "s"(np), [dend] "s"(d_end), [rsrc] "s"(rsrc), [vl] "v"(vlane16),
        [two] "n"(2 * HALF), [ring] "n"(NSLOT * HALF), [chunk] "n"(CHUNK), [simg] "n"(NIMG * CHUNK)
      : "memory", "scc");
}

__global__ __launch_bounds__(512) void mlp_loop(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grpB = wave >= 4 ? 1 : 0;
  char *WB = smem;
  const unsigned wb_lds = (unsigned)(size_t)(lptr_t)WB;         // LDS byte address of the ring
  const unsigned vlane16 = lane * 16;
  auto slot_of_half = [&](int h) { return (h % NSLOT) * HALF; };
  // buffer resource over the image (+ the repeated head chunks): base, stride 0, num_records, raw-buffer flags
  const unsigned long long ib = (unsigned long long)img;
  v4u rsrc = {(unsigned)ib, (unsigned)(ib >> 32) & 0xffffu, (unsigned)((NIMG + 4) * CHUNK), 0x00020000u};
#pragma unroll
  for (int i = 0; i < 4; ++i) rsrc[i] = __builtin_amdgcn_readfirstlane(rsrc[i]);
  // prologue: halves 0 .. PRO-1
  constexpr int PRO = NSLOT - 1 - (NSLOT == 6 ? 0 : 0);        // 6 slots: halves 0..4; 8 slots: halves 0..6
  for (int q = wave; q < (PRO / 2) * 17 + (PRO & 1) * 9; q += 8) {
    const int c = q / 17, r = q % 17;
    const int h = 2 * c + (r >= 9), off = (r >= 9 ? r - 9 : r) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(img + (size_t)c * CHUNK + r * 1024 + lane * 16), (lptr_t)(WB + slot_of_half(h) + off), 16, 0, 0);
  }
  // Stream state of an issuing wave = its pieces of the (Y, X) pair its NEXT rendezvous issues.  With LEAD = (NSLOT - 4) / 2
  // intervals of DMA flight, pair n = Y half of chunk n+LEAD (half 2n+2LEAD+1) + X half of chunk n+LEAD+1 (half 2n+2LEAD+2).
  // Group B's rendezvous at the head of chunk c >= 1 issues pair c.  Waves 4, 5: Y pieces 0-3 / 4-7; wave 6: X pieces 0-3;
  // wave 7: X pieces 4-8.
  constexpr int LEAD = (NSLOT - 4) / 2;
  const int isx = wave >= 6;                                    // this wave moves pieces of the X half
  const int p0 = (wave & 1) * 4;                                // first piece inside its half
  unsigned d = wb_lds + slot_of_half(2 * 1 + 2 * LEAD + 1 + isx) + p0 * 1024;           // pair 1 (the first one issued)
  unsigned soff = (unsigned)((1 + LEAD + isx) * CHUNK + (isx ? p0 : 9 + p0) * 1024);
  d = __builtin_amdgcn_readfirstlane(d);
  soff = __builtin_amdgcn_readfirstlane(soff);
  const unsigned d_end = __builtin_amdgcn_readfirstlane(wb_lds + NSLOT * HALF + p0 * 1024);
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
  unsigned xoff = 0;                     // ring offset of the current chunk's X half (its Y half follows at + HALF)
#ifdef LOCKSTEP
  const int goA = 1, goB = 0;
  const unsigned npA = np_w, npB = 0;
#else
  const int goA = __builtin_amdgcn_readfirstlane(1 - grpB), goB = __builtin_amdgcn_readfirstlane(grpB);
  const unsigned npA = 0, npB = np_w;
#endif
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const unsigned xnext = (xoff == (NSLOT - 2) * HALF) ? 0u : xoff + 2 * HALF;
    const char *px = WB + xoff + lane * 16;            // [bias | frags 0-7]
    const char *py = px + HALF;                         // [frags 8-15]
    const char *pnx = WB + xnext + lane * 16;           // next chunk's X half
    rendezvous(c > 0 ? goB : 0, npB, d, soff, d_end, rsrc, vlane16);      // group B: between the epilogue and X
    {
      const float b0 = *reinterpret_cast<const float *>(px - lane * 16 + (lane >> 5) * 64);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = b0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % AFD], __builtin_bit_cast(v8bf, bfrag[k]), acc0, 0, 0, 0);
      const int kn = k + AFD;
      a[k % AFD] = (kn < 8)    ? *reinterpret_cast<const v8bf *>(px + 1024 + kn * 1024)
                   : (kn < 16) ? *reinterpret_cast<const v8bf *>(py + (kn - 8) * 1024)
                               : *reinterpret_cast<const v8bf *>(pnx + 1024 + (kn - 16) * 1024);
      if (k == 7) rendezvous(goA, npA, d, soff, d_end, rsrc, vlane16);     // group A: between X and Y
      __builtin_amdgcn_sched_barrier(0);
    }
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += acc0[r];
#pragma unroll
    for (int q = 0; q < EPI; ++q) e = e * 1.0001f + 0.5f;
    sink += e * (1.0f / 65536.0f);
    __builtin_amdgcn_sched_barrier(0);
    xoff = xnext;
  }
  {
    unsigned d0 = d, s0 = soff;
    rendezvous(goB, 0, d0, s0, d_end, rsrc, vlane16);      // group B's last rendezvous (after its last epilogue)
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
  size_t lds = NSLOT * HALF;
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
