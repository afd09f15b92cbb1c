// Issue rate of v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 from one wave and from two waves of one SIMD
// (NC independent accumulator chains, straight-line): the premise of a 3-product split trunk on the 16x16x32 shape.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NC, bool BIG>
__global__ void rate(float *out, int iters, int slot) {
  v8h a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x % 64 + i)); b[i] = (_Float16)(0.002f * ((threadIdx.x % 64) * 3 + i)); }
  v4f c[NC]; v16f C[NC];
  for (int n = 0; n < NC; ++n) { for (int r = 0; r < 4; ++r) c[n][r] = 0; for (int r = 0; r < 16; ++r) C[n][r] = 0; }
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        if (BIG) C[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, C[n], 0, 0, 0);
        else c[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[n], 0, 0, 0);
      }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int n = 0; n < NC; ++n) s += BIG ? C[n][0] + C[n][7] : c[n][0] + c[n][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = (float)(t1 - t0) / (8.0f * NC * iters);
  out[64 + threadIdx.x % 8] = s;
}

int main() {
  float *out; hipMalloc(&out, 1024); hipMemset(out, 0, 1024);
  const int it = 2048;
  // one wave
  hipLaunchKernelGGL((rate<1, false>), dim3(1), dim3(64), 0, 0, out, it, 0);
  hipLaunchKernelGGL((rate<2, false>), dim3(1), dim3(64), 0, 0, out, it, 1);
  hipLaunchKernelGGL((rate<4, false>), dim3(1), dim3(64), 0, 0, out, it, 2);
  hipLaunchKernelGGL((rate<1, true>), dim3(1), dim3(64), 0, 0, out, it, 3);
  hipLaunchKernelGGL((rate<2, true>), dim3(1), dim3(64), 0, 0, out, it, 4);
  hipLaunchKernelGGL((rate<4, true>), dim3(1), dim3(64), 0, 0, out, it, 5);
  // 512 threads = 8 waves = two per SIMD
  hipLaunchKernelGGL((rate<2, false>), dim3(1), dim3(512), 0, 0, out, it, 6);
  hipLaunchKernelGGL((rate<4, false>), dim3(1), dim3(512), 0, 0, out, it, 7);
  hipLaunchKernelGGL((rate<2, true>), dim3(1), dim3(512), 0, 0, out, it, 8);
  hipLaunchKernelGGL((rate<4, true>), dim3(1), dim3(512), 0, 0, out, it, 9);
  float h[16]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("cycles per MFMA and WAVE (readcyclecounter ticks), NC = independent accumulator chains\n");
  printf("one wave       : 16x16x32  NC1 %.1f  NC2 %.1f  NC4 %.1f   |  32x32x16  NC1 %.1f  NC2 %.1f  NC4 %.1f\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  printf("two waves/SIMD : 16x16x32  NC2 %.1f  NC4 %.1f            |  32x32x16  NC2 %.1f  NC4 %.1f   (per wave: halve for the SIMD's issue interval)\n", h[6], h[7], h[8], h[9]);
  return 0;
}
