// Does hipExtAnyOrderLaunch clear the AQL barrier bit on this runtime / chip?
// (hip_ext.h says "not supported on AMD GFX9xx boards"; DESIGN.md section 3.9
// depends on the answer.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/anyorder.hip -o gpurun_out/anyorder && gpurun_out/anyorder
//
// Test 1: `waiter` (one workgroup, launched first, in order) spins on a flag
//   that only `setter` (launched behind it on the SAME stream) raises.  With
//   the barrier bit set the setter cannot start before the waiter ends, so the
//   waiter times out; with it cleared the waiter sees the flag.
// Test 2: 24 launches of a kernel that spins for a fixed time on every CU, in
//   order and any-order: wall time per launch (overlap = the launches pile up on
//   the CUs' second slots).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                          \
    }                                                                    \
  } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__global__ void waiter(unsigned* flag, unsigned* out, long long max_ticks) {
  const long long t0 = wall_clock64();  // 100 MHz
  unsigned seen = 0;
  long long t = t0;
  while (t - t0 < max_ticks) {
    if (__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      seen = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
    t = wall_clock64();
  }
  if (threadIdx.x == 0) {
    out[0] = seen;
    out[1] = (unsigned)(t - t0);
  }
}

__global__ void setter(unsigned* flag) {
  if (threadIdx.x == 0)
    __hip_atomic_store((gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void spin(long long ticks, unsigned* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (ticks < 0) sink[0] = 1;
}

int main() {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  unsigned *flag, *out;
  CHECK(hipMalloc(&flag, 4));
  CHECK(hipMalloc(&out, 8));
  for (int any = 0; any < 2; ++any) {
    CHECK(hipMemset(flag, 0, 4));
    CHECK(hipMemset(out, 0xff, 8));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, st, flag, out,
                       (long long)2000000);  // 20 ms
    hipExtLaunchKernelGGL(setter, dim3(1), dim3(64), 0, st, nullptr, nullptr,
                          any ? hipExtAnyOrderLaunch : 0, flag);
    CHECK(hipStreamSynchronize(st));
    unsigned h[2];
    CHECK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    std::printf("test1 setter %s: waiter saw the flag: %u after %.1f us\n",
                any ? "any-order" : "in-order ", h[0], h[1] / 100.0);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int any = 0; any < 2; ++any) {
      CHECK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      const int n = 24;
      for (int k = 0; k < n; ++k)
        hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 60 * 1024, st, nullptr,
                              nullptr, (any && k) ? hipExtAnyOrderLaunch : 0,
                              (long long)600 /* 6 us */, out);
      CHECK(hipStreamSynchronize(st));
      const double us =
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0)
              .count();
      std::printf("test2 %s: %d launches of a 6-us kernel (256 WGs, 60 KB LDS): %.1f us"
                  " = %.2f per launch\n",
                  any ? "any-order" : "in-order ", n, us, us / n);
    }
  return 0;
}
