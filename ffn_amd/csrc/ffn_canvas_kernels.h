// Canvas utility kernels: box reads / writes / fills, commit, the between-segment turn (integer / byte work, HBM-bound).
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

// ---------------------------------------------------------------------------
// Canvas utility kernels (integer / byte work, HBM-bound).
// ---------------------------------------------------------------------------
struct Box {
  int lo[3];
  int n[3];      // extent
  int cy, cx;    // canvas strides
};

__device__ __forceinline__ size_t box_index(const Box& b, long e) {
  const int x = e % b.n[2];
  const long t = e / b.n[2];
  const int y = t % b.n[1];
  const int z = t / b.n[1];
  return ((size_t)(b.lo[0] + z) * b.cy + (b.lo[1] + y)) * b.cx + (b.lo[2] + x);
}

template <typename T>
__global__ void box_read_kernel(const T* __restrict__ vol, Box b, long total,
                                T* __restrict__ dst) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    dst[e] = vol[box_index(b, e)];
}

template <typename T>
__global__ void box_write_kernel(T* __restrict__ vol, Box b, long total,
                                 const T* __restrict__ src) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    vol[box_index(b, e)] = src[e];
}

template <typename T>
__global__ void box_fill_kernel(T* __restrict__ vol, Box b, long total, T value) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    vol[box_index(b, e)] = value;
}

__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  // 16-byte stores, grid-stride: the per-seed "seed.clear()" (storage.py:69-71).
  const size_t n4 = n / 4;
  uint4 vv = make_uint4(v, v, v, v);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n4;
       e += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(p)[e] = vv;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = v;
}

__global__ void points_read_kernel(const float* __restrict__ seed,
                                   const int32_t* __restrict__ seg, int cz,
                                   int cy, int cx, int n,
                                   const int32_t* __restrict__ pos,
                                   float* __restrict__ seed_out,
                                   int32_t* __restrict__ seg_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int z = pos[3 * k], y = pos[3 * k + 1], x = pos[3 * k + 2];
  if (z < 0 || z >= cz || y < 0 || y >= cy || x < 0 || x >= cx) {
    seed_out[k] = __builtin_nanf("");
    seg_out[k] = 0;
    return;
  }
  const size_t ci = ((size_t)z * cy + y) * cx + x;
  seed_out[k] = seed[ci];
  seg_out[k] = seg[ci];
}

__global__ void points_write_seg_kernel(int32_t* __restrict__ seg, int cy, int cx,
                                        int n, const int32_t* __restrict__ pos,
                                        const int32_t* __restrict__ val) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  seg[((size_t)pos[3 * k] * cy + pos[3 * k + 1]) * cx + pos[3 * k + 2]] = val[k];
}

__global__ void set_seg_point_kernel(int32_t* __restrict__ seg, size_t ci,
                                     int32_t value) {
  seg[ci] = value;
}

__global__ void set_seed_point_kernel(float* __restrict__ seed, size_t ci,
                                      float value) {
  seed[ci] = value;
}

__global__ void any_segmented_kernel(const int32_t* __restrict__ seg, Box b,
                                     long total, int32_t* __restrict__ out) {
  int hit = 0;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    hit |= seg[box_index(b, e)] > 0;
  if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(out, 1);
}

// counts[0] = raw, counts[1] = actual; hist[id] += 1 for overlapped ids > 0.
__global__ void commit_count_kernel(const float* __restrict__ seed,
                                    const int32_t* __restrict__ seg, Box b,
                                    long total, float thr, int32_t max_id,
                                    unsigned long long* __restrict__ counts,
                                    unsigned* __restrict__ hist) {
  unsigned raw = 0, act = 0;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const size_t ci = box_index(b, e);
    if (seed[ci] >= thr) {  // NaN -> false
      ++raw;
      const int32_t s = seg[ci];
      if (s <= 0) {
        ++act;
      } else if (s <= max_id) {
        atomicAdd(&hist[s], 1u);
      }
    }
  }
  // wavefront reduction, then one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    raw += __shfl_xor(raw, off);
    act += __shfl_xor(act, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (raw) atomicAdd(&counts[0], (unsigned long long)raw);
    if (act) atomicAdd(&counts[1], (unsigned long long)act);
  }
}

__global__ void commit_assign_kernel(const float* __restrict__ seed,
                                     int32_t* __restrict__ seg, Box b, long total,
                                     float thr, int32_t sid) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const size_t ci = box_index(b, e);
    if (seed[ci] >= thr && seg[ci] <= 0) seg[ci] = sid;
  }
}

// ---------------------------------------------------------------------------
// The between-segment turn of Canvas.segment_all (inference.py:573-660) as ONE
// device-side sequence (ffn_canvas_segment_turn): commit count -> assign if the
// object is large enough (else the -1 marker at its seed) -> the next seeds of
// the policy tested in order (already segmented / too close to a segment, the
// latter marked -1) -> the canvas' seed volume re-initialised at the first one
// that passes.  The host reads ONE record afterwards instead of waiting for each
// answer before it queues the next kernel.
// ---------------------------------------------------------------------------
struct TurnRecord {
  unsigned long long counts[2];  // raw, actual (commit_count_kernel)
  int committed;                 // the id was assigned
  int chosen;                    // index of the next seed in the candidate list, -1 none
  int pad[2];
};
constexpr int kTurnOk = 0, kTurnSegmented = 1, kTurnTooClose = 2, kTurnNotReached = 3;

// mark_mode 0: no marker; 1: seg[mark] = -1 if it is 0 (inference.py:600-603, a
// seed that got too weak); 2: the same, but only when nothing is committed
// (inference.py:632-636, too small).
__global__ void turn_commit_kernel(const float* __restrict__ seed,
                                   int32_t* __restrict__ seg, Box b, long total,
                                   float thr, int32_t sid, long long min_size,
                                   TurnRecord* __restrict__ rec, long mark_ci,
                                   int mark_mode) {
  const bool ok = total > 0 && (long long)rec->counts[1] >= min_size;
  if (ok) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
      const size_t ci = box_index(b, e);
      if (seed[ci] >= thr && seg[ci] <= 0) seg[ci] = sid;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rec->committed = ok ? 1 : 0;
    if ((mark_mode == 1 || (mark_mode == 2 && !ok)) && seg[mark_ci] == 0)
      seg[mark_ci] = -1;
  }
}

// one wavefront per candidate: segmentation[pos] > 0 (Canvas.is_valid_pos,
// inference.py:341), else any id > 0 in the clipped box pos +- min_boundary_dist
// (inference.py:575-581)
__global__ __launch_bounds__(64) void turn_eval_kernel(
    const float* __restrict__ seed, const int32_t* __restrict__ seg, int cz, int cy,
    int cx, const int32_t* __restrict__ cand, int mz, int my, int mx,
    int* __restrict__ flags, float* __restrict__ cand_seed,
    int32_t* __restrict__ cand_seg) {
  const int j = blockIdx.x;
  const int z = cand[3 * j], y = cand[3 * j + 1], x = cand[3 * j + 2];
  const size_t ci = ((size_t)z * cy + y) * cx + x;
  const int32_t s = seg[ci];
  int flag = kTurnOk;
  if (s > 0) {
    flag = kTurnSegmented;
  } else {
    const int z0 = max(z - mz, 0), z1 = min(z + mz + 1, cz);
    const int y0 = max(y - my, 0), y1 = min(y + my + 1, cy);
    const int x0 = max(x - mx, 0), x1 = min(x + mx + 1, cx);
    const int ny = y1 - y0, nx = x1 - x0;
    const int total = (z1 - z0) * ny * nx;
    int hit = 0;
    for (int e = threadIdx.x; e < total; e += 64) {
      const int ex = e % nx, t = e / nx;
      hit |= seg[((size_t)(z0 + t / ny) * cy + (y0 + t % ny)) * cx + (x0 + ex)] > 0;
    }
    if (__any(hit)) flag = kTurnTooClose;
  }
  if (threadIdx.x == 0) {
    flags[j] = flag;
    cand_seed[j] = seed[ci];
    cand_seg[j] = s;
  }
}

// the first candidate that passed; the too-close ones BEFORE it get their -1
// (the ones after it have not been looked at as far as the caller is concerned)
__global__ __launch_bounds__(64) void turn_pick_kernel(
    int32_t* __restrict__ seg, int cy, int cx, const int32_t* __restrict__ cand,
    int n, int* __restrict__ flags, TurnRecord* __restrict__ rec) {
  int chosen = -1;
  for (int base = 0; base < n && chosen < 0; base += 64) {
    const int j = base + threadIdx.x;
    const unsigned long long m = __ballot(j < n && flags[j] == kTurnOk);
    if (m) chosen = base + __ffsll((long long)m) - 1;
  }
  const int upto = chosen < 0 ? n : chosen;
  for (int j = threadIdx.x; j < n; j += 64) {
    if (j < upto) {
      if (flags[j] == kTurnTooClose)
        seg[((size_t)cand[3 * j] * cy + cand[3 * j + 1]) * cx + cand[3 * j + 2]] = -1;
    } else if (j > upto) {
      flags[j] = kTurnNotReached;
    }
  }
  if (threadIdx.x == 0) rec->chosen = chosen;
}

// Canvas.init_seed (inference.py:282-286) at the chosen candidate: the region the
// last segment touched back to NaN, then the seed point
__global__ void turn_clear_kernel(uint32_t* __restrict__ seed, Box b, long total,
                                  int linear, size_t nvox,
                                  const TurnRecord* __restrict__ rec) {
  if (rec->chosen < 0) return;
  if (linear) {
    const size_t n4 = nvox / 4;
    const uint4 vv = make_uint4(0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u);
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n4;
         e += (size_t)gridDim.x * blockDim.x)
      reinterpret_cast<uint4*>(seed)[e] = vv;
    if (blockIdx.x == 0 && threadIdx.x < (nvox & 3))
      seed[n4 * 4 + threadIdx.x] = 0x7fc00000u;
    return;
  }
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    seed[box_index(b, e)] = 0x7fc00000u;
}

__global__ void turn_seed_kernel(float* __restrict__ seed, int cy, int cx,
                                 const int32_t* __restrict__ cand, float value,
                                 const TurnRecord* __restrict__ rec) {
  const int j = rec->chosen;
  if (j < 0) return;
  seed[((size_t)cand[3 * j] * cy + cand[3 * j + 1]) * cx + cand[3 * j + 2]] = value;
}

}  // namespace ffn
