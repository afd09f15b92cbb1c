// Micro-benchmark (round 6): the spatial trunk of the split-f16 kernels -- three partial products on v_mfma_f32_16x16x32_f16,
// weights through the 3-slot LDS-DMA chunk ring, one mid-chunk rendezvous -- in two shapes:
//   -DNW=8 -DTILES=1 : the shipped skeleton: 8 waves (two per SIMD) x one 16-sample tile = 128 samples per 17 KB chunk
//   -DNW=4 -DTILES=2 : one wave per SIMD x two 16-sample tiles: the same 128 samples per chunk, every A piece read from LDS
//                      feeds two tiles (half the fragment traffic, half the waves at the barrier), four independent accumulators
//                      per wave, activations of 32 samples (hi + lo, in + out) in 256 of the wave's 512 registers
// A layer = 8 slices x [SQ_A chunk | SQ_B chunk] (4 k-steps x 6 MFMAs per tile each); a slice's epilogue (ReLU, hi / lo split ->
// the next layer's B fragments) is real code.  -DNODMA: no weight stream.  Prints cycles per chunk and the matrix-pipe share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#ifndef NW
#define NW 8
#endif
#ifndef TILES
#define TILES 1
#endif
constexpr int CHUNK = 17 * 1024, NLAYER = 16, NIMG = 128;       /* 16 layers x 16 chunks; the image holds 128 chunks (2.2 MB) */

__device__ __forceinline__ unsigned pk(float a, float b) { v2h r = __builtin_convertvector((v2f){a, b}, v2h); return __builtin_bit_cast(unsigned, r); }
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo) {
  hi = pk(x0, x1);
  asm("" : "+v"(hi));
  const v2h hv = __builtin_bit_cast(v2h, hi);
  const _Float16 h0 = hv[0], h1 = hv[1];
  lo = pk(x0 - (float)h0, x1 - (float)h1);
}
struct Acc { v4f t0, t1; };
#ifdef ASMFRAG
/* -DASMFRAG: the A-fragment reads as inline asm, so that the compiler's counter pass does not see them (it waits for
 * lgkmcnt(0) -- every read issued so far -- in front of each fragment's first use); the waits are placed by hand with the count
 * of reads issued behind the fragment's own (LDS reads return in order) */
__device__ __forceinline__ v8h frag_asm(unsigned addr, int off) {
  v8h r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(0) : "memory");
  (void)off;
  return r;
}
template <int OFF>
__device__ __forceinline__ v8h frag_asm_o(unsigned addr) {
  v8h r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int N>
__device__ __forceinline__ void frag_wait(v8h &f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
#endif

__global__ __launch_bounds__(64 * NW) void trunk(const char *img, float *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char *WB = smem;
  int cur = 0, nxt = CHUNK, fil = 2 * CHUNK;
  /* 17 pieces over the waves: NW = 8: waves 0-4 three, wave 5 two; NW = 4: waves 0-2 four, wave 3 five */
  constexpr int PPW = NW == 8 ? 3 : 4;
  const char *src = img + wave * PPW * 1024 + lane * 16;
  int seq = 0, left = NLAYER * 16;
  auto issue = [&](int slot) {
    if (left > 0) {
#ifndef NODMA
      lptr_t dst = (lptr_t)(WB + slot + wave * PPW * 1024);
      if (NW == 8) {
        if (wave < 6) {
          __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
          if (wave < 5) __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
        }
      } else {
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)src, dst, 16, 3072, 0);
        if (wave == 3) __builtin_amdgcn_global_load_lds((gptr_t)(src + 4096), (lptr_t)((char *)dst + 4096), 16, 0, 0);
      }
#endif
      src += CHUNK;
      if (++seq == NIMG) { src -= (size_t)NIMG * CHUNK; seq = 0; }
      left -= 1;
    }
  };
  issue(cur); issue(nxt);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  v4u R0[TILES][16], R1[TILES][16];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { R0[t][e] = (v4u){0x3c003c00u + lane + t, 0x3800u + e, 0x34003400u, 0x3c00u}; R1[t][e] = (v4u){0, 0, 0, 0}; }
#ifndef NFR
#define NFR 4        /* fragment ring depth in pieces: 4 = one k-step ahead (shipped), 8 = two */
#endif
  v8h fr[NFR];
#pragma unroll
  for (int d = 0; d < NFR; ++d) fr[d] = *reinterpret_cast<const v8h *>(WB + 1024 + lane * 16 + d * 1024);
#ifdef ASMFRAG
  const unsigned lbase = (unsigned)(unsigned long long)(lptr_t)WB + 1024 + lane * 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif

  /* one chunk: 4 k-steps x [WhT0 H | WhT1 H | WlT0 H | WlT1 H | WhT0 L | WhT1 L] per tile; pieces 4 s .. 4 s + 3; HALF = 0 / 1 = SQ_A / SQ_B */
  auto chunk = [&](const v4u (&in)[TILES][16], Acc (&acc)[TILES], int half, auto &&hook) {
    const char *c = WB + cur + 1024 + lane * 16, *n = WB + nxt + 1024 + lane * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#ifdef REORDER   /* [WhT0 H | WhT1 H | WhT0 L | WhT1 L | WlT0 H | WlT1 H]: the Wh pieces are released two MFMAs earlier, every piece is re-read >= 3 reads ahead of its use */
        const int piece = j < 2 ? j : j - 2;
        const bool lo = j == 2 || j == 3;
#else
        const int piece = j < 4 ? j : j - 4;
        const bool lo = j >= 4;
#endif
#ifdef ASMFRAG
        /* first use of a piece in this k-step: j = 0..3.  Reads issued behind its own: NFR = 4: 1, 0, 3, 3; each further
         * k-step of ring depth adds 4 */
#ifdef REORDER
        if (j == 0 || j == 4 || j == 5) frag_wait<3 + (NFR - 4)>(fr[(4 * s + piece) % NFR]);
        if (j == 1) frag_wait<2 + (NFR - 4)>(fr[(4 * s + piece) % NFR]);
#else
        if (j == 0) frag_wait<1 + (NFR - 4)>(fr[(4 * s + piece) % NFR]);
        if (j == 1) frag_wait<0 + (NFR - 4)>(fr[(4 * s + piece) % NFR]);
        if (j == 2 || j == 3) frag_wait<3 + (NFR - 4)>(fr[(4 * s + piece) % NFR]);
#endif
#endif
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const v8h b = __builtin_bit_cast(v8h, in[t][2 * (4 * half + s) + (lo ? 1 : 0)]);
          if (j & 1) acc[t].t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[(4 * s + piece) % NFR], b, acc[t].t1, 0, 0, 0);
          else acc[t].t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[(4 * s + piece) % NFR], b, acc[t].t0, 0, 0, 0);
        }
        /* release: pieces 2, 3 after j = 2, 3 (their last use), pieces 0, 1 after j = 4, 5 */
#ifdef REORDER
        const int rel = j >= 2 ? j - 2 : -1;
#else
        const int rel = j == 2 ? 2 : (j == 3 ? 3 : (j == 4 ? 0 : (j == 5 ? 1 : -1)));
#endif
        if (rel >= 0) {
          const int q = 4 * s + rel + NFR;
          /* (NFR = 8: the ring runs into the next chunk from k-step 2 on -- behind the rendezvous of k-step 1, as it must) */
#ifdef ASMFRAG
          {
            const unsigned a = lbase + (q < 16 ? cur : nxt);
            v8h r;
            if (q < 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"((4 * s + rel + NFR) % 16 * 1024));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"((4 * s + rel + NFR - 16) % 16 * 1024));
            fr[(4 * s + rel) % NFR] = r;
          }
#else
          fr[(4 * s + rel) % NFR] = q < 16 ? *reinterpret_cast<const v8h *>(c + q * 1024) : *reinterpret_cast<const v8h *>(n + (q - 16) * 1024);
#endif
        }
        hook(6 * s + j);
        if (s == 1 && j == 5) {          /* mid-chunk rendezvous: chunk c + 1 landed for every wave, slot of c - 1 free */
#ifndef NOBAR
#ifdef BAREBAR
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
#endif
#endif
          issue(fil);
        }
#ifndef NOSCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    const int tmp = cur; cur = nxt; nxt = fil; fil = tmp;
  };
  auto epi_piece = [&](const Acc &a, int q, v4u &oh, v4u &ol) {
    const float y0 = q == 0 ? a.t0[0] : (q == 1 ? a.t0[2] : (q == 2 ? a.t1[0] : a.t1[2]));
    const float y1 = q == 0 ? a.t0[1] : (q == 1 ? a.t0[3] : (q == 2 ? a.t1[1] : a.t1[3]));
    const float x0 = (y0 < 0.0f) ? 0.0f : y0, x1 = (y1 < 0.0f) ? 0.0f : y1;
    unsigned hi, lo;
    split2(x0, x1, hi, lo);
    oh[q] = hi; ol[q] = lo;
  };
  auto layer = [&](const v4u (&in)[TILES][16], v4u (&outr)[TILES][16]) {
    Acc accs[2][TILES];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      Acc (&acc)[TILES] = accs[(ob + 1) & 1];
      Acc (&prev)[TILES] = accs[ob & 1];
#pragma unroll
      for (int t = 0; t < TILES; ++t) { acc[t].t0 = (v4f){0.1f, 0.2f, -0.1f, 0.05f}; acc[t].t1 = (v4f){0.0f, 0.1f, 0.2f, -0.3f}; }
      /* the previous slice's epilogue rides behind this slice's first MFMAs (TILES x 4 pieces) */
      auto hook = [&](int j) {
        if (ob == 0 || j >= 8 * TILES || (j & 1)) return;
        const int t = (j >> 1) / 4, q = (j >> 1) & 3;
        epi_piece(prev[t], q, outr[t][2 * ob - 2], outr[t][2 * ob - 1]);
      };
      chunk(in, acc, 0, hook);
      chunk(in, acc, 1, [](int) {});
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) epi_piece(accs[0][t], q, outr[t][14], outr[t][15]);
    __builtin_amdgcn_sched_barrier(0);
  };
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int l = 0; l < NLAYER / 2; ++l) {
    layer(R0, R1);
    layer(R1, R0);
  }
  const long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc ^= R0[t][e][0] ^ R0[t][e][3];
  out[blockIdx.x * 64 * NW + tid] = (float)acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  char *img; float *out; long long *cyc;
  hipMalloc(&img, (size_t)(NIMG + 2) * CHUNK); hipMemset(img, 0x11, (size_t)(NIMG + 2) * CHUNK);
  hipMalloc(&out, 256 * 64 * NW * 4); hipMalloc(&cyc, 256 * 8);
  const size_t lds = 3 * CHUNK + 90 * 1024;       /* one workgroup per CU */
  (void)hipFuncSetAttribute((const void *)trunk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(trunk, dim3(256), dim3(64 * NW), lds, 0, img, out, cyc);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(trunk, dim3(256), dim3(64 * NW), lds, 0, img, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  long long hc[256]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += hc[i] / 256.0;
  const int chunks = NLAYER * 16;
  const double mfma_cyc = 24.0 * TILES * 16.5 * (NW / 4);       /* matrix cycles per chunk and SIMD */
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)trunk);
  printf("NW=%d TILES=%d%s: %.3f ms, %.0f cycles per chunk (s_memtime), matrix issue %.0f per chunk -> %.1f %% of the pipe; %d VGPR, scratch %zu B\n", NW, TILES,
#ifdef NODMA
         " NODMA",
#else
         "",
#endif
         ms, mean / chunks, mfma_cyc, 100.0 * mfma_cyc / (mean / chunks), fa.numRegs, (size_t)fa.localSizeBytes);
  return 0;
}
