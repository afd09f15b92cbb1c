// probe: bytes per clock and CU a workgroup of 4 waves (one per SIMD) gets from an L2-resident buffer through 16-byte
// global loads -- the operand stream of the training chains.  same = 1: the four waves read the same addresses (as the
// chains do), same = 0: each wave its own quarter.  Also with the stream going through LDS (global_load_lds + ds_read_b128).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BYTES = 4 << 20;         /* streamed region per pass */
__global__ __launch_bounds__(256) void stream(const v4f *src, float *out, int passes, int same, long long *cyc) {
  __shared__ char big[100 * 1024];     /* one workgroup per CU */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = BYTES / 16;
  v4f acc = {0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    const int per = same ? n16 : n16 / 4;
    const v4f *q = src + (same ? 0 : wave * per) + lane;
#pragma unroll 1
    for (int i = 0; i < per; i += 64 * 16) {
      v4f r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = q[i + j * 64];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += r[j];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (big[threadIdx.x] == 77) acc[0] += 1.0f;
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  v4f *src; float *out; long long *cyc;
  hipMalloc(&src, BYTES); hipMemset(src, 0, BYTES); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  for (int same = 0; same < 2; ++same) {
    const int passes = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream, dim3(256), dim3(256), 0, 0, src, out, 4, same, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream, dim3(256), dim3(256), 0, 0, src, out, passes, same, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc[256]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += hc[i] / 256.0;
    const double bytes_per_cu = (double)passes * BYTES * (same ? 4 : 1);   /* bytes the CU's four waves requested */
    printf("same=%d: %.3f ms, %.1f GB/s per CU requested, %.1f TB/s chip; s_memtime cycles %.3g (100 MHz ref) -> %.1f B per core clock at 2.1 GHz\n",
           same, ms, bytes_per_cu / ms * 1e-6, bytes_per_cu * 256 / ms * 1e-9, mean, bytes_per_cu / (ms * 1e-3 * 2.1e9));
  }
  return 0;
}
