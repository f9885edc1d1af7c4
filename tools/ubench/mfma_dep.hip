// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 on one SIMD as a function of how
// many independent accumulator chains the stream alternates over (1 wave per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_dep.bin tools/ubench/mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters) {
  f16x8 a, b;
  for (int c = 0; c < 8; ++c) { a[c] = (_Float16)(threadIdx.x * 0.001f + c); b[c] = (_Float16)(c * 0.5f); }
  f32x16 acc[6];
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#define M(T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[T], 0, 0, 0);
    if (MODE == 1) { M(0) M(0) M(0) M(0) M(0) M(0) }
    if (MODE == 2) { M(0) M(1) M(0) M(1) M(0) M(1) }
    if (MODE == 3) { M(0) M(1) M(2) M(0) M(1) M(2) }
    if (MODE == 4) { M(1) M(0) M(1) M(0) M(1) M(1) }   // conv32k: C A C A C C
    if (MODE == 6) { M(0) M(1) M(2) M(3) M(4) M(5) }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  float s = 0;
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[MODE] = t1 - t0;
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 64);
  hipMemset(clk, 0, 64);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    k<1><<<256, 256>>>(out, clk, iters); k<2><<<256, 256>>>(out, clk, iters);
    k<3><<<256, 256>>>(out, clk, iters); k<4><<<256, 256>>>(out, clk, iters);
    k<6><<<256, 256>>>(out, clk, iters);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
  const char* names[8] = {"", "1 chain", "2 chains alternating", "3 chains", "C A C A C C", "", "6 chains", ""};
  for (int m : {1, 2, 3, 4, 6})
    printf("%-22s %6.2f cycles per v_mfma_f32_32x32x16_f16 (all CUs busy, 1 wave/SIMD)\n", names[m], (double)h[m] / (6.0 * iters));
  return 0;
}
