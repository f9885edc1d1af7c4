// What a kernel boundary costs on this GPU, by what the first kernel leaves behind, what the second one asks for, and WHEN the
// second launch reaches the queue.  A -> B on one stream; the gap is B's first workgroup entry minus A's last workgroup exit on
// the 100-MHz wall clock (every workgroup stores its own stamp; the host reduces).
// Build: hipcc --offload-arch=gfx950 -O2 -o boundary_probe boundary_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Fat { int pad[240]; };   // ~1 KB of kernel arguments, as the resident stack has
constexpr int kMaxBlocks = 4096;

// MODE 0 nothing, 1 plain stores, 3 sc1 write-through stores; spin_ticks > 0: stay for that long (10-ns ticks)
template <int MODE>
__global__ void __launch_bounds__(256) writer(float4* buf, int per_thread, long long* st, int spin_ticks) {
  const long long t0 = wall_clock64();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
  for (int k = 0; k < per_thread; ++k) {
    float4* p = buf + i + (size_t)k * gridDim.x * 256;
    if (MODE == 1) *p = v;
    if (MODE == 3) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)((char*)p - (char*)buf), 0, 16);
    }
  }
  if (MODE == 4) {  // a few bytes of scratch per lane (private segment), as a kernel with spills has
    volatile int arr[6];
    for (int k = 0; k < 6; ++k) arr[k] = (int)i + k;
    int acc = 0;
    for (int k = 0; k < 6; ++k) acc += arr[(k + per_thread) % 6];
    if (acc == 0x7fffffff) buf[0].x = 1.f;
  }
  if (MODE == 5) {  // 80 KB of LDS per workgroup
    extern __shared__ char a_lds[];
    a_lds[threadIdx.x] = (char)i;
    __syncthreads();
    if (a_lds[(threadIdx.x + 1) & 255] == 77 && per_thread == 99) buf[0].x = 2.f;
  }
  while (spin_ticks > 0 && wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  if (threadIdx.x == 0) st[blockIdx.x] = wall_clock64();
}

extern __shared__ char dyn_lds[];
template <bool FAT>
__global__ void __launch_bounds__(256) entry(long long* st, float4* buf, Fat fat) {
  if (threadIdx.x == 0) {
    st[kMaxBlocks + blockIdx.x] = wall_clock64();
    if (FAT && fat.pad[blockIdx.x % 240] == 12345) st[0] = 1;
  }
  if (buf && threadIdx.x == 1) dyn_lds[blockIdx.x & 63] = (char)buf[blockIdx.x].x;
}

static void busy_us(double us) {
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {}
}

template <int MODE>
static void run(const char* name, float4* buf, long long* st, int wblocks, int per_thread, int spin_us, int eblocks, int lds,
                bool fat, double host_delay_us, bool other_stream = false) {
  Fat f{};
  std::vector<long long> h(2 * kMaxBlocks);
  std::vector<double> gaps, alen;
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (lds) { CK(hipFuncSetAttribute((const void*)entry<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
             CK(hipFuncSetAttribute((const void*)entry<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
  for (int rep = 0; rep < 50; ++rep) {
    CK(hipMemsetAsync(st, 0, 2 * kMaxBlocks * sizeof(long long), s));
    CK(hipStreamSynchronize(s));
    if (MODE == 5) CK(hipFuncSetAttribute((const void*)writer<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    hipLaunchKernelGGL(writer<MODE>, dim3(wblocks), dim3(256), MODE == 5 ? 81920 : 0, s, buf, per_thread, st, spin_us * 100);
    if (host_delay_us > 0) busy_us(host_delay_us);   // B reaches the queue while A runs
    hipStream_t sb = s;
    if (other_stream) { CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(s2, ev, 0)); sb = s2; }
    if (fat) hipLaunchKernelGGL(entry<true>, dim3(eblocks), dim3(256), lds, sb, st, buf, f);
    else     hipLaunchKernelGGL(entry<false>, dim3(eblocks), dim3(256), lds, sb, st, buf, f);
    CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long a_end = 0, a_first = 0x7fffffffffffffffLL, b_in = 0x7fffffffffffffffLL;
    for (int i = 0; i < wblocks; ++i) { a_end = std::max(a_end, h[i]); a_first = std::min(a_first, h[i]); }
    for (int i = 0; i < eblocks; ++i) b_in = std::min(b_in, h[kMaxBlocks + i]);
    if (rep >= 10) { gaps.push_back((double)(b_in - a_end) * 0.01); alen.push_back((double)(a_end - a_first) * 0.01); }
  }
  std::sort(gaps.begin(), gaps.end()); std::sort(alen.begin(), alen.end());
  printf("%-52s A %4d WGs %5.1f MB spin %3d us | B %4d WGs lds %5d %-10s | host delay %5.1f us%s : gap median %5.2f us  min %5.2f  p90 %5.2f   (A first..last exit %5.1f us)\n",
         name, wblocks, wblocks * 256.0 * per_thread * 16 / 1e6, spin_us, eblocks, lds, fat ? "1KB-args" : "small-args",
         host_delay_us, other_stream ? " other stream" : "", gaps[gaps.size() / 2], gaps.front(), gaps[gaps.size() * 9 / 10], alen[alen.size() / 2]);
  CK(hipStreamDestroy(s)); CK(hipStreamDestroy(s2));
}

// The engine's stack_ahead pattern in steady state: A_i (a long kernel) runs; 60 us into it the host queues
// [B_i, A_i+1] in one burst; and so on -- the queue never drains.  Per-iteration stamps: every workgroup
// of A stores its exit, every workgroup of B its entry and exit, A_i+1 its entry.
__global__ void __launch_bounds__(256) chain_a(long long* st, int slot, int spin_ticks) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) st[slot * 2048 + blockIdx.x] = t0;              // entry
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  if (threadIdx.x == 0) st[slot * 2048 + 512 + blockIdx.x] = wall_clock64();  // exit
}
__global__ void __launch_bounds__(512) chain_b(long long* st, int slot, int spin_ticks, Fat fat) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) st[slot * 2048 + 1024 + blockIdx.x] = t0;
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  if (threadIdx.x == 0) st[slot * 2048 + 1536 + blockIdx.x] = wall_clock64();
  if (fat.pad[threadIdx.x % 240] == 12345) st[0] = 1;
}

static void chain(const char* name, int iters, double a_us, double b_us, double host_delay_us, bool burst_all) {
  long long* st; const size_t n = (size_t)(iters + 2) * 2048;
  CK(hipMalloc(&st, n * sizeof(long long))); CK(hipMemset(st, 0, n * sizeof(long long)));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Fat f{};
  CK(hipFuncSetAttribute((const void*)chain_a, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  hipLaunchKernelGGL(chain_a, dim3(356), dim3(256), 81920, s, st, 0, (int)(a_us * 100));
  for (int i = 0; i < iters; ++i) {
    if (!burst_all) {
      // wait until A_i has started (its first workgroup's entry stamp is visible), then a little more
      long long h = 0;
      while (h == 0) CK(hipMemcpy(&h, st + (size_t)i * 2048, 8, hipMemcpyDeviceToHost));
      busy_us(host_delay_us);
    }
    hipLaunchKernelGGL(chain_b, dim3(297), dim3(512), 0, s, st, i, (int)(b_us * 100), f);
    hipLaunchKernelGGL(chain_a, dim3(356), dim3(256), 81920, s, st, i + 1, (int)(a_us * 100));
  }
  CK(hipStreamSynchronize(s));
  std::vector<long long> h(n);
  CK(hipMemcpy(h.data(), st, n * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<double> g_ab, g_ba;
  for (int i = 2; i < iters; ++i) {
    long long a_end = 0, b_in = 0x7fffffffffffffffLL, b_end = 0, a_in = 0x7fffffffffffffffLL;
    for (int w = 0; w < 356; ++w) a_end = std::max(a_end, h[(size_t)i * 2048 + 512 + w]);
    for (int w = 0; w < 297; ++w) { b_in = std::min(b_in, h[(size_t)i * 2048 + 1024 + w]); b_end = std::max(b_end, h[(size_t)i * 2048 + 1536 + w]); }
    for (int w = 0; w < 356; ++w) a_in = std::min(a_in, h[(size_t)(i + 1) * 2048 + w]);
    g_ab.push_back((b_in - a_end) * 0.01); g_ba.push_back((a_in - b_end) * 0.01);
  }
  std::sort(g_ab.begin(), g_ab.end()); std::sort(g_ba.begin(), g_ba.end());
  printf("%-64s A %3.0f us, B %3.0f us: A end -> B entry %5.2f us (min %5.2f, p90 %5.2f)   B end -> next A entry %5.2f us (min %5.2f, p90 %5.2f)\n",
         name, a_us, b_us, g_ab[g_ab.size() / 2], g_ab.front(), g_ab[g_ab.size() * 9 / 10], g_ba[g_ba.size() / 2], g_ba.front(), g_ba[g_ba.size() * 9 / 10]);
  CK(hipStreamDestroy(s)); CK(hipFree(st));
}

// What hipExtLaunchKernelGGL's events measure: a 100-us kernel behind a 30-us one.
static void event_semantics() {
  long long* st; CK(hipMalloc(&st, 2 * kMaxBlocks * sizeof(long long)));
  float4* buf; CK(hipMalloc(&buf, 1 << 20));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<double> v;
    for (int rep = 0; rep < 20; ++rep) {
      hipLaunchKernelGGL(writer<0>, dim3(356), dim3(256), 0, s, buf, 0, st, 3000);
      if (mode == 0) { CK(hipEventRecord(a, s)); hipLaunchKernelGGL(writer<0>, dim3(356), dim3(256), 0, s, buf, 0, st, 10000); CK(hipEventRecord(b, s)); }
      if (mode == 1) hipExtLaunchKernelGGL(writer<0>, dim3(356), dim3(256), 0, s, a, b, 0, buf, 0, st, 10000);
      if (mode == 2) hipExtLaunchKernelGGL(writer<0>, dim3(356), dim3(256), 0, s, nullptr, c, 0, buf, 0, st, 10000);
      CK(hipStreamSynchronize(s));
      float ms = 0.f;
      hipError_t e = mode == 2 ? hipEventElapsedTime(&ms, c, c) : hipEventElapsedTime(&ms, a, b);
      if (e != hipSuccess) { printf("mode %d: hipEventElapsedTime -> %s\n", mode, hipGetErrorString(e)); (void)hipGetLastError(); break; }
      v.push_back(ms * 1e3);
    }
    if (!v.empty()) { std::sort(v.begin(), v.end());
      printf("100-us kernel behind a 30-us one, %s: %.2f us (min %.2f)\n",
             mode == 0 ? "hipEventRecord markers around it" : mode == 1 ? "hipExtLaunchKernelGGL(start, stop)" : "hipExtLaunchKernelGGL(NULL, stop), elapsed(stop, stop)",
             v[v.size() / 2], v.front()); }
  }
  CK(hipStreamDestroy(s));
}

int main() {
  event_semantics();

  printf("-- the queue never drains: [B_i, A_i+1] queued in one burst while A_i runs (hipMemcpy polls of a stamp in between)\n");
  chain("burst reaches the queue 60 us into A_i", 40, 160, 10, 60.0, false);
  chain("burst reaches the queue 5 us into A_i", 40, 160, 10, 5.0, false);
  chain("burst reaches the queue 120 us into A_i", 40, 160, 10, 120.0, false);
  chain("everything queued at once (no host in between)", 40, 160, 10, 0.0, true);

  float4* buf; long long* st;
  CK(hipMalloc(&buf, 64u << 20)); CK(hipMalloc(&st, 2 * kMaxBlocks * sizeof(long long)));
  CK(hipMemset(buf, 0, 64u << 20));
  printf("-- same burst, A stays 30 us (so B's packet is in the queue long before A ends)\n");
  run<0>("A writes nothing", buf, st, 356, 0, 30, 356, 0, false, 0);
  run<0>("A writes nothing, B 80 KB LDS", buf, st, 356, 0, 30, 356, 81920, false, 0);
  run<0>("A writes nothing, B 80 KB LDS + 1 KB args", buf, st, 356, 0, 30, 356, 81920, true, 0);
  run<1>("A plain stores 4.6 MB", buf, st, 1124, 1, 30, 356, 81920, true, 0);
  run<3>("A sc1 stores 4.6 MB", buf, st, 1124, 1, 30, 356, 81920, true, 0);
  run<1>("A plain stores 18 MB", buf, st, 1124, 4, 30, 356, 81920, true, 0);
  run<3>("A sc1 stores 18 MB", buf, st, 1124, 4, 30, 356, 81920, true, 0);
  run<1>("A plain stores 0.15 MB (a paste)", buf, st, 36, 1, 30, 356, 81920, true, 0);
  printf("-- B reaches the queue while A runs (A stays 200 us)\n");
  for (double d : {0.0, 20.0, 60.0, 120.0, 170.0})
    run<0>("A writes nothing", buf, st, 356, 0, 200, 356, 81920, true, d);
  run<3>("A sc1 stores 4.6 MB", buf, st, 1124, 1, 200, 356, 81920, true, 60.0);
  run<1>("A plain stores 4.6 MB", buf, st, 1124, 1, 200, 356, 81920, true, 60.0);
  run<0>("A writes nothing, B small", buf, st, 356, 0, 200, 8, 0, false, 60.0);
  run<0>("A writes nothing, B on another stream (event)", buf, st, 356, 0, 200, 356, 81920, true, 60.0, true);
  run<0>("A writes nothing, B on another stream (event)", buf, st, 356, 0, 200, 356, 81920, true, 0.0, true);
  printf("-- what A is: scratch, LDS (B reaches the queue 60 us into A's 200)\n");
  run<4>("A uses scratch", buf, st, 356, 0, 200, 297, 43008, true, 60.0);
  run<4>("A uses scratch, same burst", buf, st, 356, 0, 200, 297, 43008, true, 0.0);
  run<5>("A has 80 KB LDS", buf, st, 356, 0, 200, 297, 43008, true, 60.0);
  run<5>("A has 80 KB LDS, same burst", buf, st, 356, 0, 200, 297, 43008, true, 0.0);
  run<0>("plain A, B 297 WGs x 43 KB", buf, st, 356, 0, 200, 297, 43008, true, 60.0);
  printf("-- B arrives after A has ended (A 5 us, B 40 us later): the idle queue's launch latency, for scale\n");
  run<0>("A writes nothing", buf, st, 356, 0, 5, 356, 81920, true, 40.0);
  return 0;
}
