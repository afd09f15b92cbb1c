// probe: does v_cvt_pk_f16_f32 (gfx950) agree with v_cvt_f16_f32 for every fp32 input?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(unsigned long long *cnt, unsigned *first) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const unsigned bits = (unsigned)i;
    const float x = __builtin_bit_cast(float, bits);
    unsigned pk, single;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(x));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(single) : "v"(x));
    const unsigned a = pk & 0xffff, b = single & 0xffff;
    const bool nan_both = ((a & 0x7c00) == 0x7c00 && (a & 0x3ff)) && ((b & 0x7c00) == 0x7c00 && (b & 0x3ff));
    if (a != b && !nan_both) {
      const unsigned long long n = atomicAdd(cnt, 1ull);
      if (n < 16) { first[2 * n] = bits; first[2 * n + 1] = (a << 16) | b; }
    }
  }
}
int main() {
  unsigned long long *dc, hc = 0; unsigned *df, hf[32] = {0};
  hipMalloc(&dc, 8); hipMalloc(&df, sizeof(hf)); hipMemset(dc, 0, 8); hipMemset(df, 0, sizeof(hf));
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, dc, df);
  hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
  printf("mismatches between v_cvt_pk_f16_f32 and v_cvt_f16_f32 over all 2^32 inputs: %llu\n", hc);
  for (int i = 0; i < 16 && i < (int)hc; ++i) { float x = __builtin_bit_cast(float, hf[2 * i]); printf("  x = %.9g (0x%08x): pk 0x%04x  single 0x%04x\n", x, hf[2 * i], hf[2 * i + 1] >> 16, hf[2 * i + 1] & 0xffff); }
  return 0;
}
