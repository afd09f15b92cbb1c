// probe (round 6): what bounds the weight stream of the level kernels -- every CU pulling the SAME 3.5 MB image from its XCD's L2
// into a 3-slot LDS ring by LDS-DMA (global_load_lds_dwordx4), 17 x 1 KB pieces per chunk, one rendezvous per chunk.
// Nothing but the stream runs here.  Variants (argv[1]):
//   0 chunk-contiguous image (the shipped layout): piece q of chunk c at c * 17 KB + q * 1 KB
//   1 piece-major image: piece q of chunk c at q * (n_chunks KB + pad) + c * 1 KB  (a chunk's pieces 200 KB apart: other L2 channels)
//   2 shipped layout, every workgroup starts at its own chunk (blockIdx * 37 % n_chunks): are the CUs' synchronous requests the bound?
//   3 shipped layout, lookahead of TWO chunks (4-slot ring, rendezvous waits for the older chunk only)
//   4 private image per XCD-local slot: each workgroup streams its own 3.5 MB copy (no sharing at all; 256 x 3.5 MB = 900 MB: L2 misses)
// Prints TB/s chip-wide and B/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
constexpr int CHUNK = 17 * 1024, NCH = 207, SLOTS = 4;

template <int MODE>
__global__ __launch_bounds__(512) void stream(const char *img, int passes, long long *cyc, float *sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long t0 = __builtin_readcyclecounter();
  const size_t pstride = (size_t)NCH * 1024 + 4096 + 256;        /* piece-major: + an odd number of 256-byte units */
  const char *base = img + (MODE == 4 ? (size_t)blockIdx.x * ((size_t)NCH * CHUNK) : 0);
  int c0 = MODE == 2 ? (int)((blockIdx.x * 37u) % NCH) : 0;
  int issued = 0, slot = 0;
  const int total = passes * NCH;
  auto issue = [&](int k) {        /* chunk number k of the stream into ring slot `slot` */
    const int c = (c0 + k) % NCH;
    if (wave < 6) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int q = 3 * wave + j < 17 ? 3 * wave + j : 16;     /* (wave 5 repeats piece 16: three pieces per wave, a uniform vmcnt) */
        {
          const char *src = MODE == 1 ? base + (size_t)q * pstride + (size_t)c * 1024 + lane * 16
                                      : base + (size_t)c * CHUNK + q * 1024 + lane * 16;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + slot * CHUNK + q * 1024), 16, 0, 0);
        }
      }
    }
    slot = (slot + 1) % SLOTS;
  };
  constexpr int AHEAD = MODE == 3 ? 3 : 2;      /* chunks in flight / landed ahead of the consumer */
  for (; issued < AHEAD && issued < total; ++issued) issue(issued);
  float acc = 0.0f;
  for (int k = 0; k < total; ++k) {
    /* rendezvous of chunk k: it has landed for every wave */
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   /* the two newer chunks (<= 3 pieces each) may still fly */
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __syncthreads();
    if (issued < total) { issue(issued); ++issued; }
    acc += *reinterpret_cast<const float *>(lds + (k % SLOTS) * CHUNK + threadIdx.x * 4);    /* one LDS read per chunk: the data is used */
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
static void run(const char *img, int passes, long long *cyc, float *sink, const char *what) {
  const size_t ldsb = SLOTS * CHUNK + 60 * 1024;   /* > 80 KB: one workgroup per CU */
  hipFuncSetAttribute((const void *)stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(stream<MODE>, dim3(256), dim3(512), ldsb, 0, img, 2, cyc, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL(stream<MODE>, dim3(256), dim3(512), ldsb, 0, img, passes, cyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * passes * NCH * CHUNK;
  printf("mode %d (%s): %.3f ms, %.2f TB/s chip, %.1f GB/s per CU, %.1f B/clk/CU at 2.1 GHz, %.0f ns per 17 KB chunk\n", MODE, what, ms,
         bytes / ms * 1e-9, bytes / 256 / ms * 1e-6, bytes / 256 / (ms * 1e-3 * 2.1e9), ms * 1e6 / (passes * NCH));
}

int main(int argc, char **argv) {
  const int passes = argc > 1 ? atoi(argv[1]) : 40;
  char *img; long long *cyc; float *sink;
  const size_t big = (size_t)256 * NCH * CHUNK + (1 << 20);
  hipMalloc(&img, big); hipMemset(img, 0, big); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
  run<0>(img, passes, cyc, sink, "chunk-contiguous image, all CUs in step");
  run<1>(img, passes, cyc, sink, "piece-major image");
  run<2>(img, passes, cyc, sink, "chunk-contiguous, staggered start per workgroup");
  run<3>(img, passes, cyc, sink, "chunk-contiguous, two chunks of lookahead");
  run<4>(img, passes, cyc, sink, "a private image per workgroup (no sharing)");
  run<0>(img, passes, cyc, sink, "chunk-contiguous again");
  return 0;
}
