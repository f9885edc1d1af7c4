// Which load flavour reads another XCD's sc1 write-through stores FRESH when the
// reader's own L2 (and L1) hold the line's previous contents?  (The resident
// conv stack re-reads the same activation buffers every second layer.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_handoff.hip -o xcd_handoff
//
// Grid = 16 workgroups (b -> XCD b % 8 by observation; XCC_ID is recorded).
// Workgroup 0 produces, workgroup `cons` consumes (1: another XCD; 8: the same
// XCD, another CU).  Round i: the producer stores i into a 16-KB buffer (16-B
// stores of flavour S), drains, publishes i; the consumer polls, reads the
// buffer with flavour L (registers, or LDS-DMA + ds_read), counts words != i,
// acknowledges.  Every round re-reads what the last round left in its caches.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                          \
    }                                                                    \
  } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWords = 4096;  // 16 KB
constexpr long long kTimeout = 20000000;  // 200 ms of the 100 MHz clock

template <int S>
__device__ __forceinline__ void store16(unsigned* p, unsigned off, u32x4 v) {
  if (S == 0) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(p) : "memory");
  if (S == 1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(p) : "memory");
  if (S == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(p) : "memory");
}

template <int L>
__device__ __forceinline__ u32x4 load16(const unsigned* p, unsigned off) {
  u32x4 v;
  if (L == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(p) : "memory");
  if (L == 1) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(p) : "memory");
  if (L == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off), "s"(p) : "memory");
  if (L == 3) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(p) : "memory");
  if (L == 4) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(v) : "v"(off), "s"(p) : "memory");
  if (L == 5) asm volatile("global_load_dwordx4 %0, %1, %2 sc1 nt" : "=v"(v) : "v"(off), "s"(p) : "memory");
  return v;
}

template <int L>
__device__ __forceinline__ void dma16(const unsigned* p, unsigned off, unsigned lds) {
  if (L == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(p), "s"(lds) : "memory");
  if (L == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(off), "s"(p), "s"(lds) : "memory");
  if (L == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1" ::"v"(off), "s"(p), "s"(lds) : "memory");
  if (L == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(off), "s"(p), "s"(lds) : "memory");
  if (L == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0" ::"v"(off), "s"(p), "s"(lds) : "memory");
  if (L == 5) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1 nt" ::"v"(off), "s"(p), "s"(lds) : "memory");
}

__device__ __forceinline__ bool wait_for(unsigned* f, unsigned want) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load((gu32*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > kTimeout) return false;
  }
  return true;
}

// MODE 0: register loads; 1: LDS-DMA; 2: register loads behind buffer_inv sc1
template <int S, int L, int MODE>
__global__ __launch_bounds__(256) void handoff(unsigned* buf, unsigned* flags, int cons,
                                               int rounds, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[kWords];
  const int tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (tid == 0) out[8 + blockIdx.x] = xcc;
  unsigned* ready = flags;       // producer -> consumer
  unsigned* ack = flags + 64;    // consumer -> producer
  if (blockIdx.x == 0) {
    for (int i = 1; i <= rounds; ++i) {
      const u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
#pragma unroll
      for (int k = 0; k < kWords / 4 / 256; ++k)
        store16<S>(buf, (unsigned)(tid + 256 * k) * 16, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store((gu32*)ready, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_for(ack, (unsigned)i)) out[2] = 1;
      }
      __syncthreads();
    }
  } else if ((int)blockIdx.x == cons) {
    unsigned stale = 0;
    long long t_sum = 0;
    for (int i = 1; i <= rounds; ++i) {
      if (tid == 0 && !wait_for(ready, (unsigned)i)) out[3] = 1;
      __syncthreads();
      const long long t0 = wall_clock64();
      if (MODE == 2) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
      }
      if (MODE == 1) {
        const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
#pragma unroll
        for (int k = 0; k < kWords / 4 / 256; ++k)
          dma16<L>(buf, (unsigned)(tid + 256 * k) * 16,
                   __builtin_amdgcn_readfirstlane(lbase + ((tid >> 6) * 64 + 256 * k) * 16));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWords / 4 / 256; ++k) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&lds[(tid + 256 * k) * 4]);
          stale += (v[0] != (unsigned)i) + (v[1] != (unsigned)i) + (v[2] != (unsigned)i) + (v[3] != (unsigned)i);
        }
      } else {
        u32x4 v[kWords / 4 / 256];
#pragma unroll
        for (int k = 0; k < kWords / 4 / 256; ++k) v[k] = load16<L>(buf, (unsigned)(tid + 256 * k) * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < kWords / 4 / 256; ++k) {
          asm volatile("" : "+v"(v[k]));
          stale += (v[k][0] != (unsigned)i) + (v[k][1] != (unsigned)i) + (v[k][2] != (unsigned)i) + (v[k][3] != (unsigned)i);
        }
      }
      t_sum += wall_clock64() - t0;
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store((gu32*)ack, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    atomicAdd(&out[0], stale);
    if (tid == 0) out[1] = (unsigned)t_sum;
  }
}

template <int S, int L, int MODE>
int run(const char* name, unsigned* buf, unsigned* flags, unsigned* out, int cons) {
  const int rounds = 2000;
  CHECK(hipMemset(buf, 0, kWords * 4));
  CHECK(hipMemset(flags, 0, 1024));
  CHECK(hipMemset(out, 0, 256));
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((handoff<S, L, MODE>), dim3(16), dim3(256), 0, 0, buf, flags, cons, rounds, out);
  CHECK(hipDeviceSynchronize());
  unsigned h[64];
  CHECK(hipMemcpy(h, out, 256, hipMemcpyDeviceToHost));
  std::printf("%-44s consumer wg %2d (xcc %u, producer xcc %u): stale words %9u of %d  read %.2f us/round%s%s\n",
              name, cons, h[8 + cons], h[8], h[0], rounds * kWords, h[1] / 100.0 / rounds,
              h[2] ? " PRODUCER-TIMEOUT" : "", h[3] ? " CONSUMER-TIMEOUT" : "");
  return 0;
}

int main() {
  unsigned *buf, *flags, *out;
  CHECK(hipMalloc(&buf, kWords * 4));
  CHECK(hipMalloc(&flags, 1024));
  CHECK(hipMalloc(&out, 256));
  for (int cons : {1, 8}) {
#define RUN(S, L, MODE, NAME) if (run<S, L, MODE>(NAME, buf, flags, out, cons)) return 1
    RUN(1, 0, 0, "store sc1     | load plain      (registers)");
    RUN(1, 4, 0, "store sc1     | load sc0        (registers)");
    RUN(1, 1, 0, "store sc1     | load sc1        (registers)");
    RUN(1, 2, 0, "store sc1     | load sc0 sc1    (registers)");
    RUN(1, 3, 0, "store sc1     | load nt         (registers)");
    RUN(1, 5, 0, "store sc1     | load sc1 nt     (registers)");
    RUN(2, 1, 0, "store sc0 sc1 | load sc1        (registers)");
    RUN(2, 2, 0, "store sc0 sc1 | load sc0 sc1    (registers)");
    RUN(0, 1, 0, "store plain   | load sc1        (registers)");
    RUN(0, 2, 0, "store plain   | load sc0 sc1    (registers)");
    RUN(1, 0, 2, "store sc1     | buffer_inv sc1 + plain loads ");
    RUN(1, 0, 1, "store sc1     | load plain      (LDS-DMA)");
    RUN(1, 1, 1, "store sc1     | load sc1        (LDS-DMA)");
    RUN(1, 2, 1, "store sc1     | load sc0 sc1    (LDS-DMA)");
    RUN(1, 3, 1, "store sc1     | load nt         (LDS-DMA)");
    RUN(1, 5, 1, "store sc1     | load sc1 nt     (LDS-DMA)");
    RUN(2, 2, 1, "store sc0 sc1 | load sc0 sc1    (LDS-DMA)");
    RUN(0, 1, 1, "store plain   | load sc1        (LDS-DMA)");
  }
  return 0;
}
