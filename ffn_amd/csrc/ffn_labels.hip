// libffn_hip.so -- label operations declared in include/ffn_labels.h.
//
// All kernels here are HBM-bound integer streaming kernels (gfx950): joint-id
// histograms through a two-level (LDS, then global) open-addressing hash table,
// table-driven relabelling, and union-find connected components.  No MFMA, no
// float: the roofline is HBM bytes (ffn_labels_last_timing).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/ffn_hip.h"
#include "../../include/ffn_labels.h"
#include "ffn_internal.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 kEmptyKey = ~0ull;
constexpr u32 kBackground = 0xffffffffu;
constexpr int kThreads = 256;
constexpr int kLdsSlots = 2048;      // per-block pre-aggregation table
constexpr int kLdsProbes = 16;
constexpr u32 kMaxProbes = 1u << 14;  // global table: give up -> grow + retry
constexpr int kScanTile = 2048;       // elements per block in the root ranking

__device__ __forceinline__ u32 mix64(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (u32)k;
}

// Slot of `key` in the global table, inserting it if absent.
__device__ __forceinline__ u32 table_insert(u64* keys, u32 mask, u64 key,
                                            int* overflow) {
  u32 slot = mix64(key) & mask;
  for (u32 probe = 0; probe < kMaxProbes; ++probe) {
    u64 prev = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    if (prev == kEmptyKey) prev = atomicCAS(&keys[slot], kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) return slot;
    slot = (slot + 1) & mask;
    // table already known to be too small: stop probing, the host regrows it
    if ((probe & 255) == 255 &&
        __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return kBackground;
  }
  *overflow = 1;
  return kBackground;
}

// Slot of `key` (read only); kBackground if absent.
__device__ __forceinline__ u32 table_find(const u64* keys, u32 mask, u64 key) {
  u32 slot = mix64(key) & mask;
  for (u32 probe = 0; probe < kMaxProbes; ++probe) {
    const u64 k = keys[slot];
    if (k == key) return slot;
    if (k == kEmptyKey) return kBackground;
    slot = (slot + 1) & mask;
  }
  return kBackground;
}

template <typename T>
__device__ __forceinline__ u64 pair_key(const T* a, const T* b, size_t i) {
  const u64 ka = (u64)a[i];
  return b ? (ka | ((u64)b[i] << 32)) : ka;
}

// Lanes holding the same key as their left neighbour form a run; only the
// first lane of a run (the leader) touches a hash table.  Returns the leader
// mask; `valid` lanes must form a prefix of the wave.
__device__ __forceinline__ u64 run_leaders(u64 key, bool valid, int lane) {
  const u64 left = __shfl_up(key, 1);
  return __ballot(valid && (lane == 0 || left != key));
}

template <typename T>
__global__ __launch_bounds__(kThreads) void pair_count_kernel(
    const T* __restrict__ a, const T* __restrict__ b, size_t n, u64* keys,
    u64* counts, u32 mask, int* overflow) {
  __shared__ u64 skeys[kLdsSlots];
  __shared__ u32 scnt[kLdsSlots];
  for (int s = threadIdx.x; s < kLdsSlots; s += kThreads) {
    skeys[s] = kEmptyKey;
    scnt[s] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // Each block owns one contiguous span (good pre-aggregation: neighbouring
  // voxels share ids), swept in coalesced rows of kThreads voxels.
  const size_t per_block =
      ((n + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;
  const size_t lo = (size_t)blockIdx.x * per_block;
  const size_t hi = lo + per_block < n ? lo + per_block : n;
  for (size_t base = lo; base < hi; base += kThreads) {
    const size_t i = base + threadIdx.x;
    const bool valid = i < hi;
    const u64 key = valid ? pair_key(a, b, i) : kEmptyKey;
    if (valid && (key == kEmptyKey ||
                  (sizeof(T) == 8 && b && (((u64)a[i] | (u64)b[i]) >> 32))))
      *overflow = 2;  // id outside the packable range (caller must remap)
    const u64 leaders = run_leaders(key, valid, lane);
    const int nvalid = __popcll(__ballot(valid));
    if (valid && ((leaders >> lane) & 1)) {
      const u64 above = lane == 63 ? 0 : leaders & ~((2ull << lane) - 1);
      const int end = above ? __ffsll((long long)above) - 1 : nvalid;
      const u32 run = (u32)(end - lane);
      u32 s = mix64(key) & (kLdsSlots - 1);
      bool done = false;
      for (int probe = 0; probe < kLdsProbes; ++probe) {
        const u64 prev = atomicCAS(&skeys[s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
          atomicAdd(&scnt[s], run);
          done = true;
          break;
        }
        s = (s + 1) & (kLdsSlots - 1);
      }
      if (!done) {  // block table crowded: straight to the global one
        const u32 g = table_insert(keys, mask, key, overflow);
        if (g != kBackground) atomicAdd(&counts[g], (u64)run);
      }
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < kLdsSlots; s += kThreads) {
    const u32 c = scnt[s];
    if (c) {
      const u32 g = table_insert(keys, mask, skeys[s], overflow);
      if (g != kBackground) atomicAdd(&counts[g], (u64)c);
    }
  }
}

__global__ void table_compact_kernel(const u64* keys, const u64* counts,
                                     u32 nslots, u64* out_key, u64* out_count,
                                     u32* out_slot, u32 cap, u32* n_out) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const u64 k = keys[s];
  if (k == kEmptyKey) return;
  const u32 j = atomicAdd(n_out, 1u);
  if (j < cap) {
    out_key[j] = k;
    out_count[j] = counts[s];
    out_slot[j] = s;
  }
}

__global__ void scatter_labels_kernel(const u32* slot, const u64* label,
                                      u32 n, u64* slot_label) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) slot_label[slot[k]] = label[k];
}

// out[i] = slot_label[slot of (a[i], b[i])]; `missing` < 0: every key is
// present (pair labels); 0: absent -> 0; 1: absent -> the key itself (remap).
template <typename T>
__global__ __launch_bounds__(kThreads) void table_apply_kernel(
    const T* __restrict__ a, const T* __restrict__ b, size_t n,
    const u64* __restrict__ keys, const u64* __restrict__ slot_label, u32 mask,
    int missing, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * kThreads;
  const size_t rounds = (n + stride - 1) / stride;
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  for (size_t r = 0; r < rounds; ++r, i += stride) {
    const bool valid = i < n;
    const u64 key = valid ? pair_key(a, b, i) : kEmptyKey;
    const u64 leaders = run_leaders(key, valid, lane);
    u64 label = 0;
    if (valid && ((leaders >> lane) & 1)) {
      const u32 s = table_find(keys, mask, key);
      label = s != kBackground ? slot_label[s] : (missing > 0 ? key : 0);
    }
    const u64 below = leaders & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
    const int src = below ? 63 - __clzll((long long)below) : lane;
    label = __shfl(label, src);
    if (valid) out[i] = (T)label;
  }
}

__global__ void map_build_kernel(const u64* in_keys, const u64* in_vals, u32 n,
                                 u64* keys, u64* vals, u32 mask,
                                 int* overflow) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u32 s = table_insert(keys, mask, in_keys[k], overflow);
  if (s != kBackground) vals[s] = in_vals[k];
}

// ---- connected components: union-find over voxel indices -------------------

__device__ __forceinline__ u32 uf_load(const u32* parent, u32 i) {
  return __hip_atomic_load(&parent[i], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u32 uf_find(const u32* parent, u32 i) {
  u32 p;
  while ((p = uf_load(parent, i)) != i) i = p;
  return i;
}

// Links the larger root under the smaller one: the root of a finished
// component is its smallest flat index = its first voxel in raster order.
__device__ __forceinline__ void uf_union(u32* parent, u32 x, u32 y) {
  bool done = false;
  while (!done) {
    x = uf_find(parent, x);
    y = uf_find(parent, y);
    if (x < y) {
      const u32 old = atomicMin(&parent[y], x);
      done = old == y;
      y = old;
    } else if (y < x) {
      const u32 old = atomicMin(&parent[x], y);
      done = old == x;
      x = old;
    } else {
      done = true;
    }
  }
}

// parent[i] = start of the x-run of equal labels i belongs to within its wave
// row segment (cheap pre-linking: most unions along x never reach the atomics).
template <typename T>
__global__ __launch_bounds__(kThreads) void cc_init_kernel(
    const T* __restrict__ in, u32 n, u32 nx, u32* parent, u32* first_zero) {
  const u32 i = blockIdx.x * kThreads + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < n;
  const u64 v = valid ? (u64)in[i] : 0;
  const u32 x = valid ? i % nx : 0;
  const u64 left = __shfl_up(v, 1);
  const bool leader = !valid || lane == 0 || x == 0 || left != v;
  const u64 leaders = __ballot(leader);
  const u64 below = leaders & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
  const int src = 63 - __clzll((long long)below);  // lane 0 is always a leader
  if (!valid) return;
  if (v == 0) {
    parent[i] = kBackground;
    // one hot address: look before the atomic (the minimum settles after the
    // first few workgroups, later ones only read)
    if (leader && i < __hip_atomic_load(first_zero, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(first_zero, i);
  } else {
    parent[i] = i - (u32)(lane - src);
  }
}

struct CcGeom {
  u32 nz, ny, nx;
  int n_off;
  int off[13][3];
};

template <typename T>
__global__ __launch_bounds__(kThreads) void cc_merge_kernel(
    const T* __restrict__ in, u32 n, CcGeom g, u32* parent) {
  const u32 i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const T v = in[i];
  if (v == 0) return;
  const u32 x = i % g.nx;
  const u32 y = (i / g.nx) % g.ny;
  const u32 z = i / (g.nx * g.ny);
  for (int k = 0; k < g.n_off; ++k) {
    const int zz = (int)z + g.off[k][0];
    const int yy = (int)y + g.off[k][1];
    const int xx = (int)x + g.off[k][2];
    if (zz < 0 || yy < 0 || xx < 0 || zz >= (int)g.nz || yy >= (int)g.ny ||
        xx >= (int)g.nx)
      continue;
    const u32 j = ((u32)zz * g.ny + (u32)yy) * g.nx + (u32)xx;
    if (in[j] == v) {
      // Two x-runs that touch along a stretch need ONE union: skip the pair
      // (i, j) when (i-1, j-1) joins the same two runs.
      if (x > 0 && xx > 0 && in[i - 1] == v && in[j - 1] == v &&
          (g.off[k][0] != 0 || g.off[k][1] != 0))
        continue;
      // skip the atomics when i is pre-linked to j through its x-run
      if (uf_load(parent, i) == j || uf_load(parent, j) == uf_load(parent, i))
        continue;
      uf_union(parent, i, j);
    }
  }
}

__global__ __launch_bounds__(kThreads) void cc_flatten_kernel(u32 n,
                                                              u32* parent) {
  const u32 i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const u32 p = parent[i];
  if (p == kBackground || p == i) return;
  parent[i] = uf_find(parent, p);
}

// Roots per tile of kScanTile voxels.
__global__ __launch_bounds__(kThreads) void cc_count_roots_kernel(
    const u32* __restrict__ parent, u32 n, u32* tile_roots) {
  __shared__ u32 wsum[kThreads / 64];
  const u32 base = blockIdx.x * kScanTile;
  u32 mine = 0;
  for (int k = 0; k < kScanTile / kThreads; ++k) {
    const u32 i = base + k * kThreads + threadIdx.x;
    mine += (i < n && parent[i] == i) ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += wsum[w];
    tile_roots[blockIdx.x] = t;
  }
}

// Exclusive scan of tile_roots in place (one block; ntiles <= 2^21).
__global__ __launch_bounds__(1024) void cc_scan_tiles_kernel(u32* tile_roots,
                                                             u32 ntiles,
                                                             u32* total) {
  __shared__ u32 part[1024];
  const u32 per = (ntiles + 1023) / 1024;
  const u32 lo = threadIdx.x * per;
  const u32 hi = lo + per < ntiles ? lo + per : ntiles;
  u32 s = 0;
  for (u32 t = lo; t < hi; ++t) s += tile_roots[t];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (int t = 0; t < 1024; ++t) {
      const u32 v = part[t];
      part[t] = run;
      run += v;
    }
    *total = run;
  }
  __syncthreads();
  u32 run = part[threadIdx.x];
  for (u32 t = lo; t < hi; ++t) {
    const u32 v = tile_roots[t];
    tile_roots[t] = run;
    run += v;
  }
}

// newid[root] = 1 + number of roots before it (raster order).
__global__ __launch_bounds__(kThreads) void cc_rank_roots_kernel(
    const u32* __restrict__ parent, u32 n, const u32* __restrict__ tile_base,
    u32* newid, u64* first_index, u32 cap) {
  __shared__ u32 wcount[kThreads / 64];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = tile_base[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32 base = blockIdx.x * kScanTile;
  for (int k = 0; k < kScanTile / kThreads; ++k) {
    const u32 i = base + k * kThreads + threadIdx.x;
    const bool root = i < n && parent[i] == i;
    const u64 m = __ballot(root);
    if (lane == 0) wcount[wave] = (u32)__popcll(m);
    __syncthreads();
    u32 before = carry;
    for (int w = 0; w < wave; ++w) before += wcount[w];
    if (root) {
      const u32 id = before + (u32)__popcll(m & ((1ull << lane) - 1));
      newid[i] = id + 1;
      if (first_index && id < cap) first_index[id] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 t = 0;
      for (int w = 0; w < kThreads / 64; ++w) t += wcount[w];
      carry += t;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void cc_output_kernel(
    const u32* __restrict__ parent, const u32* __restrict__ newid, u32 n,
    T* __restrict__ out, u64* sizes, u32 cap) {
  const u32 i = blockIdx.x * kThreads + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < n;
  const u32 p = valid ? parent[i] : kBackground;
  const u32 id = p == kBackground ? 0u : newid[p];
  if (valid) out[i] = (T)id;
  if (sizes) {
    // one atomic per run of equal ids in the wave
    const u64 leaders = run_leaders((u64)id, valid, lane);
    const int nvalid = __popcll(__ballot(valid));
    if (valid && id != 0 && ((leaders >> lane) & 1)) {
      const u64 above = lane == 63 ? 0 : leaders & ~((2ull << lane) - 1);
      const int end = above ? __ffsll((long long)above) - 1 : nvalid;
      if (id - 1 < cap) atomicAdd(&sizes[id - 1], (u64)(end - lane));
    }
  }
}

// ---- device-resident assembly of sub-box results -----------------------------

struct Box3 {
  long long lo[3], hi[3];     // core of the sub-box, sub-box coordinates
  long long src_shape[3];
  long long dst_shape[3];
  long long corner[3];        // the sub-box inside the assembled volume
};

// dst[n] = src[n] > 0 ? src[n] : 0 (the -1 "excluded" markers of a canvas)
__global__ __launch_bounds__(kThreads) void copy_labels_kernel(
    const int32_t* __restrict__ src, size_t n, int32_t* __restrict__ dst) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const int32_t v = src[i];
    dst[i] = v > 0 ? v : 0;
  }
}

// assembled[corner + v] = src[v] > 0 ? src[v] + off : 0 over the core; one block
// row of threads per x-run (coalesced on both sides)
__global__ __launch_bounds__(kThreads) void place_core_kernel(
    const int32_t* __restrict__ src, int32_t off, int32_t* __restrict__ dst,
    Box3 b) {
  const long long nx = b.hi[2] - b.lo[2], ny = b.hi[1] - b.lo[1];
  const long long rows = (b.hi[0] - b.lo[0]) * ny;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long z = b.lo[0] + r / ny, y = b.lo[1] + r % ny;
    const int32_t* s = src + (z * b.src_shape[1] + y) * b.src_shape[2] + b.lo[2];
    int32_t* d = dst + ((z + b.corner[0]) * b.dst_shape[1] + (y + b.corner[1])) *
                           b.dst_shape[2] + b.corner[2] + b.lo[2];
    for (long long x = threadIdx.x; x < nx; x += kThreads) {
      const int32_t v = s[x];
      d[x] = v > 0 ? v + off : 0;
    }
  }
}

// the two label volumes of a sub-box's MARGIN (everything outside its core):
// a = own label + off, b = assembled label; both 0 inside the core
__global__ __launch_bounds__(kThreads) void margin_gather_kernel(
    const int32_t* __restrict__ own, int32_t off,
    const int32_t* __restrict__ assembled, u32* __restrict__ a,
    u32* __restrict__ bb, Box3 b) {
  const long long nx = b.src_shape[2], ny = b.src_shape[1];
  const long long rows = b.src_shape[0] * ny;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long long z = r / ny, y = r % ny;
    const bool row_in_core = z >= b.lo[0] && z < b.hi[0] && y >= b.lo[1] &&
                             y < b.hi[1];
    const int32_t* s = own + r * nx;
    const int32_t* g = assembled + ((z + b.corner[0]) * b.dst_shape[1] +
                                    (y + b.corner[1])) * b.dst_shape[2] +
                       b.corner[2];
    for (long long x = threadIdx.x; x < nx; x += kThreads) {
      const bool core = row_in_core && x >= b.lo[2] && x < b.hi[2];
      const int32_t v = s[x];
      a[r * nx + x] = core ? 0u : (v > 0 ? (u32)(v + off) : 0u);
      bb[r * nx + x] = core ? 0u : (u32)g[x];
    }
  }
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct ffn_labels {
  int device_id = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf a, b, out, keys, counts, slot_label, aux0, aux1, aux2, small;
  u32 nslots = 0;  // size of the resident hash table (pair_counts)
  size_t n = 0;    // voxels of the resident volumes
  int elem_bytes = 0;
  bool have_b = false;
  bool pairs_valid = false;
  double last_ms = 0.0, last_bytes = 0.0;
};

namespace {

#define L_TRY(expr)                                                           \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess)                                                     \
      return ffn_set_error(FFN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,       \
                           hipGetErrorString(_e), __FILE__, __LINE__);        \
  } while (0)

int ensure(DevBuf& buf, size_t bytes) {
  if (buf.bytes >= bytes && buf.p) return FFN_OK;
  if (buf.p) L_TRY(hipFree(buf.p));
  buf.p = nullptr;
  buf.bytes = 0;
  L_TRY(hipMalloc(&buf.p, bytes ? bytes : 16));
  buf.bytes = bytes ? bytes : 16;
  return FFN_OK;
}

#define L_OK(expr)                \
  do {                            \
    int _rc = (expr);             \
    if (_rc != FFN_OK) return _rc; \
  } while (0)

int grid_for(size_t n, int per_block) {
  return (int)std::min<size_t>((n + per_block - 1) / per_block, 1u << 30);
}

int start_timer(ffn_labels* h) {
  L_TRY(hipEventRecord(h->ev0, h->stream));
  return FFN_OK;
}

int stop_timer(ffn_labels* h, double bytes) {
  L_TRY(hipEventRecord(h->ev1, h->stream));
  L_TRY(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  L_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  h->last_bytes = bytes;
  return FFN_OK;
}

u32 table_size_for(size_t expected) {
  u32 s = 1u << 16;
  while (s < (1u << 30) && (size_t)s < expected * 2) s <<= 1;
  return s;
}

int alloc_table(ffn_labels* h, u32 nslots, bool with_counts) {
  L_OK(ensure(h->keys, (size_t)nslots * 8));
  L_OK(ensure(h->slot_label, (size_t)nslots * 8));
  if (with_counts) L_OK(ensure(h->counts, (size_t)nslots * 8));
  L_TRY(hipMemsetAsync(h->keys.p, 0xff, (size_t)nslots * 8, h->stream));
  if (with_counts)
    L_TRY(hipMemsetAsync(h->counts.p, 0, (size_t)nslots * 8, h->stream));
  return FFN_OK;
}

template <typename T>
int pair_counts_impl(ffn_labels* h, size_t n, size_t cap, uint64_t* pair_a,
                     uint64_t* pair_b, uint64_t* pair_count,
                     uint32_t* pair_slot, size_t* n_pairs) {
  const T* a = static_cast<const T*>(h->a.p);
  const T* b = h->have_b ? static_cast<const T*>(h->b.p) : nullptr;
  L_OK(ensure(h->small, 64));
  int* overflow = static_cast<int*>(h->small.p);
  u32* n_out = reinterpret_cast<u32*>(h->small.p) + 1;
  u32 nslots = std::max<u32>(h->nslots, 1u << 20);
  for (;;) {
    L_OK(alloc_table(h, nslots, true));
    L_TRY(hipMemsetAsync(h->small.p, 0, 64, h->stream));
    L_OK(start_timer(h));
    const int blocks = (int)std::min<size_t>(
        2048, std::max<size_t>(1, (n + 16 * kThreads - 1) / (16 * kThreads)));
    hipLaunchKernelGGL((pair_count_kernel<T>), dim3(blocks), dim3(kThreads), 0,
                       h->stream, a, b, n, static_cast<u64*>(h->keys.p),
                       static_cast<u64*>(h->counts.p), nslots - 1, overflow);
    L_TRY(hipGetLastError());
    L_OK(stop_timer(h, (double)n * sizeof(T) * (b ? 2 : 1)));
    int ov = 0;
    L_TRY(hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost,
                         h->stream));
    L_TRY(hipStreamSynchronize(h->stream));
    if (!ov) break;
    if (ov == 2)
      return ffn_set_error(FFN_ERR_ARG,
                           "label id >= 2^32 - 1: remap ids before pairing");
    if (nslots >= (1u << 30))
      return ffn_set_error(FFN_ERR_ARG, "pair table overflow at 2^30 slots");
    nslots <<= 2;
  }
  h->nslots = nslots;
  // compact the occupied slots into dense arrays (order unspecified)
  const size_t want = std::min<size_t>(cap, nslots);
  L_OK(ensure(h->aux0, want * 8));
  L_OK(ensure(h->aux1, want * 8));
  L_OK(ensure(h->aux2, want * 4));
  hipLaunchKernelGGL(table_compact_kernel, dim3((nslots + 255) / 256),
                     dim3(256), 0, h->stream,
                     static_cast<const u64*>(h->keys.p),
                     static_cast<const u64*>(h->counts.p), nslots,
                     static_cast<u64*>(h->aux0.p), static_cast<u64*>(h->aux1.p),
                     static_cast<u32*>(h->aux2.p), (u32)want, n_out);
  L_TRY(hipGetLastError());
  u32 found = 0;
  L_TRY(hipMemcpyAsync(&found, n_out, sizeof(u32), hipMemcpyDeviceToHost,
                       h->stream));
  L_TRY(hipStreamSynchronize(h->stream));
  *n_pairs = found;
  if (found > cap)
    return ffn_set_error(FFN_ERR_ARG, "%u unique pairs exceed cap %zu", found,
                         cap);
  std::vector<u64> keys(found);
  L_TRY(hipMemcpy(keys.data(), h->aux0.p, (size_t)found * 8,
                  hipMemcpyDeviceToHost));
  L_TRY(hipMemcpy(pair_count, h->aux1.p, (size_t)found * 8,
                  hipMemcpyDeviceToHost));
  L_TRY(hipMemcpy(pair_slot, h->aux2.p, (size_t)found * 4,
                  hipMemcpyDeviceToHost));
  for (u32 k = 0; k < found; ++k) {
    pair_a[k] = h->have_b ? (keys[k] & 0xffffffffull) : keys[k];
    pair_b[k] = h->have_b ? (keys[k] >> 32) : 0;
  }
  h->pairs_valid = true;
  return FFN_OK;
}

template <typename T>
int apply_impl(ffn_labels* h, const T* a, const T* b, size_t n, int missing,
               T* out) {
  const int blocks =
      (int)std::min<size_t>(4096, std::max<size_t>(1, (n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL((table_apply_kernel<T>), dim3(blocks), dim3(kThreads), 0,
                     h->stream, a, b, n, static_cast<const u64*>(h->keys.p),
                     static_cast<const u64*>(h->slot_label.p), h->nslots - 1,
                     missing, out);
  L_TRY(hipGetLastError());
  return FFN_OK;
}

template <typename T>
int cc_impl(ffn_labels* h, u32 n, const CcGeom& g, size_t cap,
            uint64_t* n_components, bool want_first, bool want_sizes) {
  const T* in = static_cast<const T*>(h->a.p);
  T* out = static_cast<T*>(h->out.p);
  u32* parent = static_cast<u32*>(h->aux0.p);
  u32* newid = static_cast<u32*>(h->aux1.p);
  const u32 ntiles = (n + kScanTile - 1) / kScanTile;
  u32* tiles = static_cast<u32*>(h->aux2.p);
  u32* first_zero = reinterpret_cast<u32*>(h->small.p);
  u32* total = first_zero + 1;
  u64* first_index = want_first ? static_cast<u64*>(h->keys.p) : nullptr;
  u64* sizes = want_sizes ? static_cast<u64*>(h->counts.p) : nullptr;
  const int blocks = (int)((n + kThreads - 1) / kThreads);
  L_TRY(hipMemsetAsync(h->small.p, 0xff, 4, h->stream));
  if (sizes) L_TRY(hipMemsetAsync(sizes, 0, cap * 8, h->stream));
  L_OK(start_timer(h));
  hipLaunchKernelGGL((cc_init_kernel<T>), dim3(blocks), dim3(kThreads), 0,
                     h->stream, in, n, g.nx, parent, first_zero);
  hipLaunchKernelGGL((cc_merge_kernel<T>), dim3(blocks), dim3(kThreads), 0,
                     h->stream, in, n, g, parent);
  hipLaunchKernelGGL(cc_flatten_kernel, dim3(blocks), dim3(kThreads), 0,
                     h->stream, n, parent);
  hipLaunchKernelGGL(cc_count_roots_kernel, dim3(ntiles), dim3(kThreads), 0,
                     h->stream, parent, n, tiles);
  hipLaunchKernelGGL(cc_scan_tiles_kernel, dim3(1), dim3(1024), 0, h->stream,
                     tiles, ntiles, total);
  hipLaunchKernelGGL(cc_rank_roots_kernel, dim3(ntiles), dim3(kThreads), 0,
                     h->stream, parent, n, tiles, newid, first_index, (u32)cap);
  hipLaunchKernelGGL((cc_output_kernel<T>), dim3(blocks), dim3(kThreads), 0,
                     h->stream, parent, newid, n, out, sizes, (u32)cap);
  L_TRY(hipGetLastError());
  // in read by init + merge (neighbour reads hit L2), parent/newid traffic,
  // out written: 2 * sizeof(T) + 5 * 4 bytes per voxel is the streaming floor.
  L_OK(stop_timer(h, (double)n * (2 * sizeof(T) + 20)));
  u32 host[2] = {0, 0};
  L_TRY(hipMemcpy(host, h->small.p, 8, hipMemcpyDeviceToHost));
  *n_components = host[1];
  h->last_bytes = (double)n * (2 * sizeof(T) + 20);
  // host[0] = first zero index (0xffffffff if none); returned by the caller
  return (int)FFN_OK;
}

}  // namespace

extern "C" {

int ffn_labels_create(int device_id, ffn_labels** out) {
  if (!out) return ffn_set_error(FFN_ERR_ARG, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  L_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev)
    return ffn_set_error(FFN_ERR_ARG, "device %d not present (%d devices)",
                         device_id, ndev);
  L_TRY(hipSetDevice(device_id));
  ffn_labels* h = new ffn_labels();
  h->device_id = device_id;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  if (e != hipSuccess) {
    ffn_labels_destroy(h);
    return ffn_set_error(FFN_ERR_HIP, "stream/event creation failed: %s",
                         hipGetErrorString(e));
  }
  *out = h;
  return FFN_OK;
}

void ffn_labels_destroy(ffn_labels* h) {
  if (!h) return;
  (void)hipSetDevice(h->device_id);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (DevBuf* b : {&h->a, &h->b, &h->out, &h->keys, &h->counts,
                    &h->slot_label, &h->aux0, &h->aux1, &h->aux2, &h->small})
    if (b->p) (void)hipFree(b->p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int ffn_labels_pair_counts(ffn_labels* h, const void* a, const void* b,
                           int elem_bytes, size_t n, size_t cap,
                           uint64_t* pair_a, uint64_t* pair_b,
                           uint64_t* pair_count, uint32_t* pair_slot,
                           size_t* n_pairs) {
  if (!h || !a || !pair_a || !pair_b || !pair_count || !pair_slot || !n_pairs)
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (elem_bytes != 4 && elem_bytes != 8)
    return ffn_set_error(FFN_ERR_ARG, "elem_bytes must be 4 or 8");
  L_TRY(hipSetDevice(h->device_id));
  h->pairs_valid = false;
  *n_pairs = 0;
  h->n = n;
  h->elem_bytes = elem_bytes;
  h->have_b = b != nullptr;
  if (n == 0) {
    h->nslots = std::max<u32>(h->nslots, 1u << 20);
    L_OK(alloc_table(h, h->nslots, true));
    L_TRY(hipStreamSynchronize(h->stream));
    h->pairs_valid = true;
    h->last_ms = 0.0;
    h->last_bytes = 0.0;
    return FFN_OK;
  }
  L_OK(ensure(h->a, n * elem_bytes));
  L_TRY(hipMemcpyAsync(h->a.p, a, n * elem_bytes, hipMemcpyHostToDevice,
                       h->stream));
  if (b) {
    L_OK(ensure(h->b, n * elem_bytes));
    L_TRY(hipMemcpyAsync(h->b.p, b, n * elem_bytes, hipMemcpyHostToDevice,
                         h->stream));
  }
  if (elem_bytes == 4)
    return pair_counts_impl<uint32_t>(h, n, cap, pair_a, pair_b, pair_count,
                                      pair_slot, n_pairs);
  return pair_counts_impl<uint64_t>(h, n, cap, pair_a, pair_b, pair_count,
                                    pair_slot, n_pairs);
}

int ffn_labels_apply_pair_labels(ffn_labels* h, size_t n_pairs,
                                 const uint32_t* pair_slot,
                                 const uint64_t* new_label, void* out) {
  if (!h || (n_pairs && (!pair_slot || !new_label)) || (h && h->n && !out))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (!h->pairs_valid)
    return ffn_set_error(FFN_ERR_STATE,
                         "no pair table resident: call ffn_labels_pair_counts");
  L_TRY(hipSetDevice(h->device_id));
  if (h->n == 0) return FFN_OK;
  for (size_t k = 0; k < n_pairs; ++k)
    if (pair_slot[k] >= h->nslots)
      return ffn_set_error(FFN_ERR_ARG, "pair_slot[%zu] out of range", k);
  L_OK(ensure(h->aux2, n_pairs * 4));
  L_OK(ensure(h->aux1, n_pairs * 8));
  L_TRY(hipMemcpyAsync(h->aux2.p, pair_slot, n_pairs * 4,
                       hipMemcpyHostToDevice, h->stream));
  L_TRY(hipMemcpyAsync(h->aux1.p, new_label, n_pairs * 8,
                       hipMemcpyHostToDevice, h->stream));
  L_OK(ensure(h->out, h->n * h->elem_bytes));
  L_OK(start_timer(h));
  if (n_pairs)
    hipLaunchKernelGGL(scatter_labels_kernel, dim3((n_pairs + 255) / 256),
                       dim3(256), 0, h->stream,
                       static_cast<const u32*>(h->aux2.p),
                       static_cast<const u64*>(h->aux1.p), (u32)n_pairs,
                       static_cast<u64*>(h->slot_label.p));
  if (h->elem_bytes == 4)
    L_OK(apply_impl<uint32_t>(
        h, static_cast<const uint32_t*>(h->a.p),
        h->have_b ? static_cast<const uint32_t*>(h->b.p) : nullptr, h->n, -1,
        static_cast<uint32_t*>(h->out.p)));
  else
    L_OK(apply_impl<uint64_t>(
        h, static_cast<const uint64_t*>(h->a.p),
        h->have_b ? static_cast<const uint64_t*>(h->b.p) : nullptr, h->n, -1,
        static_cast<uint64_t*>(h->out.p)));
  L_OK(stop_timer(h, (double)h->n * h->elem_bytes * (h->have_b ? 3 : 2)));
  L_TRY(hipMemcpy(out, h->out.p, h->n * h->elem_bytes, hipMemcpyDeviceToHost));
  return FFN_OK;
}

int ffn_labels_remap(ffn_labels* h, const void* in, int elem_bytes, size_t n,
                     size_t n_keys, const uint64_t* keys,
                     const uint64_t* values, int keep_missing, void* out) {
  if (!h || (n && (!in || !out)) || (n_keys && (!keys || !values)))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (elem_bytes != 4 && elem_bytes != 8)
    return ffn_set_error(FFN_ERR_ARG, "elem_bytes must be 4 or 8");
  if (n_keys >= (1u << 29))
    return ffn_set_error(FFN_ERR_ARG, "too many keys");
  L_TRY(hipSetDevice(h->device_id));
  h->pairs_valid = false;
  if (n == 0) return FFN_OK;
  L_OK(ensure(h->small, 64));
  int* overflow = static_cast<int*>(h->small.p);
  u32 nslots = table_size_for(n_keys);
  L_OK(ensure(h->aux0, std::max<size_t>(n_keys, 1) * 8));
  L_OK(ensure(h->aux1, std::max<size_t>(n_keys, 1) * 8));
  L_TRY(hipMemcpyAsync(h->aux0.p, keys, n_keys * 8, hipMemcpyHostToDevice,
                       h->stream));
  L_TRY(hipMemcpyAsync(h->aux1.p, values, n_keys * 8, hipMemcpyHostToDevice,
                       h->stream));
  for (;;) {
    L_OK(alloc_table(h, nslots, false));
    L_TRY(hipMemsetAsync(h->small.p, 0, 64, h->stream));
    if (n_keys)
      hipLaunchKernelGGL(map_build_kernel, dim3((n_keys + 255) / 256),
                         dim3(256), 0, h->stream,
                         static_cast<const u64*>(h->aux0.p),
                         static_cast<const u64*>(h->aux1.p), (u32)n_keys,
                         static_cast<u64*>(h->keys.p),
                         static_cast<u64*>(h->slot_label.p), nslots - 1,
                         overflow);
    L_TRY(hipGetLastError());
    int ov = 0;
    L_TRY(hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost,
                         h->stream));
    L_TRY(hipStreamSynchronize(h->stream));
    if (!ov) break;
    if (nslots >= (1u << 30))
      return ffn_set_error(FFN_ERR_ARG, "remap table overflow");
    nslots <<= 2;
  }
  h->nslots = nslots;
  h->n = n;
  h->elem_bytes = elem_bytes;
  h->have_b = false;
  L_OK(ensure(h->a, n * elem_bytes));
  L_OK(ensure(h->out, n * elem_bytes));
  L_TRY(hipMemcpyAsync(h->a.p, in, n * elem_bytes, hipMemcpyHostToDevice,
                       h->stream));
  L_OK(start_timer(h));
  if (elem_bytes == 4)
    L_OK(apply_impl<uint32_t>(h, static_cast<const uint32_t*>(h->a.p), nullptr,
                              n, keep_missing ? 1 : 0,
                              static_cast<uint32_t*>(h->out.p)));
  else
    L_OK(apply_impl<uint64_t>(h, static_cast<const uint64_t*>(h->a.p), nullptr,
                              n, keep_missing ? 1 : 0,
                              static_cast<uint64_t*>(h->out.p)));
  L_OK(stop_timer(h, (double)n * elem_bytes * 2));
  L_TRY(hipMemcpy(out, h->out.p, n * elem_bytes, hipMemcpyDeviceToHost));
  return FFN_OK;
}

int ffn_labels_connected_components(ffn_labels* h, const void* in,
                                    int elem_bytes, const int64_t shape_zyx[3],
                                    int connectivity, void* out,
                                    uint64_t* n_components, size_t cap,
                                    uint64_t* first_index, uint64_t* sizes,
                                    int64_t* first_zero_index) {
  if (!h || !shape_zyx || !n_components)
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (elem_bytes != 4 && elem_bytes != 8)
    return ffn_set_error(FFN_ERR_ARG, "elem_bytes must be 4 or 8");
  if (connectivity < 1 || connectivity > 3)
    return ffn_set_error(FFN_ERR_ARG, "connectivity must be 1, 2 or 3");
  for (int k = 0; k < 3; ++k)
    if (shape_zyx[k] < 0)
      return ffn_set_error(FFN_ERR_ARG, "negative shape");
  const double nd =
      (double)shape_zyx[0] * (double)shape_zyx[1] * (double)shape_zyx[2];
  if (nd >= 4294967295.0)
    return ffn_set_error(FFN_ERR_ARG, "volume must have < 2^32 - 1 voxels");
  const u32 n = (u32)(shape_zyx[0] * shape_zyx[1] * shape_zyx[2]);
  *n_components = 0;
  if (first_zero_index) *first_zero_index = -1;
  L_TRY(hipSetDevice(h->device_id));
  h->pairs_valid = false;
  if (n == 0) return FFN_OK;
  if (!in || !out) return ffn_set_error(FFN_ERR_ARG, "NULL volume");
  CcGeom g;
  g.nz = (u32)shape_zyx[0];
  g.ny = (u32)shape_zyx[1];
  g.nx = (u32)shape_zyx[2];
  g.n_off = 0;
  // the 13 "earlier in raster order" neighbours, filtered by connectivity
  for (int dz = -1; dz <= 0; ++dz)
    for (int dy = -1; dy <= (dz < 0 ? 1 : 0); ++dy)
      for (int dx = -1; dx <= ((dz < 0 || dy < 0) ? 1 : -1); ++dx) {
        const int order = (dz != 0) + (dy != 0) + (dx != 0);
        if (order == 0 || order > connectivity) continue;
        g.off[g.n_off][0] = dz;
        g.off[g.n_off][1] = dy;
        g.off[g.n_off][2] = dx;
        ++g.n_off;
      }
  const u32 ntiles = (n + kScanTile - 1) / kScanTile;
  L_OK(ensure(h->a, (size_t)n * elem_bytes));
  L_OK(ensure(h->out, (size_t)n * elem_bytes));
  L_OK(ensure(h->aux0, (size_t)n * 4));
  L_OK(ensure(h->aux1, (size_t)n * 4));
  L_OK(ensure(h->aux2, (size_t)ntiles * 4));
  L_OK(ensure(h->small, 64));
  if (first_index) L_OK(ensure(h->keys, std::max<size_t>(cap, 1) * 8));
  if (sizes) L_OK(ensure(h->counts, std::max<size_t>(cap, 1) * 8));
  L_TRY(hipMemcpyAsync(h->a.p, in, (size_t)n * elem_bytes,
                       hipMemcpyHostToDevice, h->stream));
  if (elem_bytes == 4)
    L_OK(cc_impl<uint32_t>(h, n, g, cap, n_components, first_index != nullptr,
                           sizes != nullptr));
  else
    L_OK(cc_impl<uint64_t>(h, n, g, cap, n_components, first_index != nullptr,
                           sizes != nullptr));
  u32 fz = 0;
  L_TRY(hipMemcpy(&fz, h->small.p, 4, hipMemcpyDeviceToHost));
  if (first_zero_index) *first_zero_index = fz == kBackground ? -1 : (int64_t)fz;
  L_TRY(hipMemcpy(out, h->out.p, (size_t)n * elem_bytes,
                  hipMemcpyDeviceToHost));
  const size_t ncopy = std::min<size_t>(cap, *n_components);
  if (first_index && ncopy)
    L_TRY(hipMemcpy(first_index, h->keys.p, ncopy * 8, hipMemcpyDeviceToHost));
  if (sizes && ncopy)
    L_TRY(hipMemcpy(sizes, h->counts.p, ncopy * 8, hipMemcpyDeviceToHost));
  if ((first_index || sizes) && *n_components > cap)
    return ffn_set_error(FFN_ERR_ARG, "%llu components exceed cap %zu",
                         (unsigned long long)*n_components, cap);
  return FFN_OK;
}

namespace {
int fill_box(Box3* b, const int64_t src_shape[3], const int64_t core_lo[3],
             const int64_t core_hi[3], const int64_t dst_shape[3],
             const int64_t corner[3]) {
  for (int k = 0; k < 3; ++k) {
    b->lo[k] = core_lo[k];
    b->hi[k] = core_hi[k];
    b->src_shape[k] = src_shape[k];
    b->dst_shape[k] = dst_shape[k];
    b->corner[k] = corner[k];
    if (core_lo[k] < 0 || core_hi[k] > src_shape[k] || core_lo[k] > core_hi[k] ||
        corner[k] < 0 || corner[k] + src_shape[k] > dst_shape[k])
      return ffn_set_error(FFN_ERR_ARG, "sub-box / core outside its volume (axis %d)", k);
  }
  return FFN_OK;
}
}  // namespace

int ffn_labels_copy_device(ffn_labels* h, const int32_t* src_dev, size_t n,
                           int32_t* dst_dev) {
  if (!h || (n && (!src_dev || !dst_dev)))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  L_TRY(hipSetDevice(h->device_id));
  if (n)
    hipLaunchKernelGGL(copy_labels_kernel, dim3(grid_for(n, kThreads * 8)),
                       dim3(kThreads), 0, h->stream, src_dev, n, dst_dev);
  L_TRY(hipGetLastError());
  L_TRY(hipStreamSynchronize(h->stream));
  return FFN_OK;
}

int ffn_labels_copy_canvas(ffn_labels* h, ffn_canvas* canvas, int32_t* dst_dev) {
  if (!h || !canvas || !dst_dev) return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  FfnCanvasView v;
  L_OK(ffn_canvas_view(canvas, &v));
  if (v.device_id != h->device_id)
    return ffn_set_error(FFN_ERR_ARG, "canvas lives on device %d, labels on %d",
                         v.device_id, h->device_id);
  L_TRY(hipSetDevice(h->device_id));
  // the canvas' own stream may still be committing the last segment
  L_TRY(hipStreamSynchronize(static_cast<hipStream_t>(v.engine_stream)));
  const size_t n = (size_t)v.shape_zyx[0] * v.shape_zyx[1] * v.shape_zyx[2];
  return ffn_labels_copy_device(h, v.segmentation, n, dst_dev);
}

int ffn_labels_place_core_device(ffn_labels* h, const int32_t* src_dev,
                                 const int64_t src_shape_zyx[3],
                                 const int64_t core_lo[3], const int64_t core_hi[3],
                                 int32_t id_offset, int32_t* dst_dev,
                                 const int64_t dst_shape_zyx[3],
                                 const int64_t corner_zyx[3]) {
  if (!h || !src_dev || !dst_dev || !src_shape_zyx || !core_lo || !core_hi ||
      !dst_shape_zyx || !corner_zyx)
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  L_TRY(hipSetDevice(h->device_id));
  Box3 b;
  L_OK(fill_box(&b, src_shape_zyx, core_lo, core_hi, dst_shape_zyx, corner_zyx));
  const long long rows = (b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]);
  if (rows > 0 && b.hi[2] > b.lo[2])
    hipLaunchKernelGGL(place_core_kernel,
                       dim3((unsigned)std::min<long long>(rows, 1 << 20)),
                       dim3(kThreads), 0, h->stream, src_dev, id_offset, dst_dev, b);
  L_TRY(hipGetLastError());
  L_TRY(hipStreamSynchronize(h->stream));
  return FFN_OK;
}

int ffn_labels_margin_pairs_device(ffn_labels* h, const int32_t* own_dev,
                                   const int64_t own_shape_zyx[3],
                                   int32_t id_offset, const int64_t core_lo[3],
                                   const int64_t core_hi[3],
                                   const int32_t* assembled_dev,
                                   const int64_t assembled_shape_zyx[3],
                                   const int64_t corner_zyx[3], size_t cap,
                                   uint64_t* pair_a, uint64_t* pair_b,
                                   uint64_t* pair_count, size_t* n_pairs) {
  if (!h || !own_dev || !assembled_dev || !own_shape_zyx || !core_lo ||
      !core_hi || !assembled_shape_zyx || !corner_zyx || !pair_a || !pair_b ||
      !pair_count || !n_pairs)
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  L_TRY(hipSetDevice(h->device_id));
  Box3 b;
  L_OK(fill_box(&b, own_shape_zyx, core_lo, core_hi, assembled_shape_zyx,
                corner_zyx));
  const size_t n = (size_t)own_shape_zyx[0] * own_shape_zyx[1] * own_shape_zyx[2];
  h->pairs_valid = false;
  *n_pairs = 0;
  if (n == 0) return FFN_OK;
  h->n = n;
  h->elem_bytes = 4;
  h->have_b = true;
  L_OK(ensure(h->a, n * 4));
  L_OK(ensure(h->b, n * 4));
  const long long rows = own_shape_zyx[0] * own_shape_zyx[1];
  hipLaunchKernelGGL(margin_gather_kernel,
                     dim3((unsigned)std::min<long long>(rows, 1 << 20)),
                     dim3(kThreads), 0, h->stream, own_dev, id_offset,
                     assembled_dev, static_cast<u32*>(h->a.p),
                     static_cast<u32*>(h->b.p), b);
  L_TRY(hipGetLastError());
  std::vector<uint32_t> slots(cap ? cap : 1);
  return pair_counts_impl<uint32_t>(h, n, cap, pair_a, pair_b, pair_count,
                                    slots.data(), n_pairs);
}

int ffn_labels_remap_device(ffn_labels* h, int32_t* vol_dev, size_t n,
                            size_t n_keys, const uint64_t* keys,
                            const uint64_t* values) {
  if (!h || (n && !vol_dev) || (n_keys && (!keys || !values)))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (n_keys >= (1u << 29)) return ffn_set_error(FFN_ERR_ARG, "too many keys");
  L_TRY(hipSetDevice(h->device_id));
  h->pairs_valid = false;
  if (n == 0 || n_keys == 0) return FFN_OK;
  L_OK(ensure(h->small, 64));
  int* overflow = static_cast<int*>(h->small.p);
  u32 nslots = table_size_for(n_keys);
  L_OK(ensure(h->aux0, n_keys * 8));
  L_OK(ensure(h->aux1, n_keys * 8));
  L_TRY(hipMemcpyAsync(h->aux0.p, keys, n_keys * 8, hipMemcpyHostToDevice,
                       h->stream));
  L_TRY(hipMemcpyAsync(h->aux1.p, values, n_keys * 8, hipMemcpyHostToDevice,
                       h->stream));
  for (;;) {
    L_OK(alloc_table(h, nslots, false));
    L_TRY(hipMemsetAsync(h->small.p, 0, 64, h->stream));
    hipLaunchKernelGGL(map_build_kernel, dim3((n_keys + 255) / 256), dim3(256), 0,
                       h->stream, static_cast<const u64*>(h->aux0.p),
                       static_cast<const u64*>(h->aux1.p), (u32)n_keys,
                       static_cast<u64*>(h->keys.p),
                       static_cast<u64*>(h->slot_label.p), nslots - 1, overflow);
    L_TRY(hipGetLastError());
    int ov = 0;
    L_TRY(hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost,
                         h->stream));
    L_TRY(hipStreamSynchronize(h->stream));
    if (!ov) break;
    if (nslots >= (1u << 30))
      return ffn_set_error(FFN_ERR_ARG, "remap table overflow");
    nslots <<= 2;
  }
  h->nslots = nslots;
  h->n = n;
  h->elem_bytes = 4;
  h->have_b = false;
  L_OK(start_timer(h));
  // in place: every voxel reads its own label and writes its own slot
  L_OK(apply_impl<uint32_t>(h, reinterpret_cast<const uint32_t*>(vol_dev), nullptr,
                            n, 1, reinterpret_cast<uint32_t*>(vol_dev)));
  L_OK(stop_timer(h, (double)n * 4 * 2));
  return FFN_OK;
}

int ffn_labels_last_timing(ffn_labels* h, double* kernel_ms,
                           double* algorithmic_bytes) {
  if (!h) return ffn_set_error(FFN_ERR_ARG, "NULL handle");
  if (kernel_ms) *kernel_ms = h->last_ms;
  if (algorithmic_bytes) *algorithmic_bytes = h->last_bytes;
  return FFN_OK;
}

}  // extern "C"
