// libffn_hip.so -- PolicyPeaks seed generation declared in include/ffn_seeds.h.
//
// A chain of HBM-bound streaming kernels (gfx950).  Floating-point steps mirror
// scipy's arithmetic exactly (see the header): f64 accumulation in
// correlate1d's order, one f32 rounding per separable pass, and NO fused
// multiply-add anywhere in this file.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/ffn_seeds.h"
#include "ffn_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRadius = 255;

struct Shape {
  int nz, ny, nx;
};

__device__ __forceinline__ int reflect_index(int i, int n) {
  // scipy mode='reflect' (d c b a | a b c d | d c b a), any distance
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - 1 - i : i;
}

__device__ __forceinline__ void decode(size_t i, const Shape& s, int& z, int& y,
                                       int& x) {
  x = (int)(i % s.nx);
  const size_t r = i / s.nx;
  y = (int)(r % s.ny);
  z = (int)(r / s.ny);
}

__device__ __forceinline__ size_t neighbour(const Shape& s, int axis, int z,
                                            int y, int x, int pos) {
  if (axis == 0) z = pos;
  if (axis == 1) y = pos;
  if (axis == 2) x = pos;
  return ((size_t)z * s.ny + y) * s.nx + x;
}

// out = f32( in*wc + (left +- right)*ws ), accumulated in f64 like
// NI_Correlate1D's symmetric / anti-symmetric branches (3 taps).
__global__ __launch_bounds__(kThreads) void tap3_kernel(
    const float* __restrict__ in, float* __restrict__ out, Shape s, size_t n,
    int axis, double wc, double ws, int anti) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int z, y, x;
  decode(i, s, z, y, x);
  const int len = axis == 0 ? s.nz : axis == 1 ? s.ny : s.nx;
  const int l = axis == 0 ? z : axis == 1 ? y : x;
  const double c = (double)in[i];
  const double a = (double)in[neighbour(s, axis, z, y, x,
                                        reflect_index(l - 1, len))];
  const double b = (double)in[neighbour(s, axis, z, y, x,
                                        reflect_index(l + 1, len))];
  double t = c * wc;
  const double pair = anti ? (a - b) : (a + b);
  const double prod = pair * ws;
  t = t + prod;
  out[i] = (float)t;
}

// acc = first ? d*d : acc + d*d   (f32, numpy.multiply / +=); last: sqrt
__global__ __launch_bounds__(kThreads) void square_acc_kernel(
    const float* __restrict__ d, float* __restrict__ acc, size_t n, int first,
    int last) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float v = d[i];
  const float sq = v * v;
  float a = first ? sq : acc[i] + sq;
  // correctly rounded f32 sqrt (numpy's): v_sqrt_f32 is 1 ulp, so go through
  // f64 (53 >= 2*24+2 bits: the double rounding is innocuous)
  if (last) a = (float)__dsqrt_rn((double)a);
  acc[i] = a;
}

__global__ __launch_bounds__(kThreads) void gauss_kernel(
    const float* __restrict__ in, float* __restrict__ out, Shape s, size_t n,
    int axis, const double* __restrict__ weights, int radius) {
  __shared__ double w[2 * kMaxRadius + 1];
  for (int k = threadIdx.x; k < 2 * radius + 1; k += kThreads)
    w[k] = weights[k];
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int z, y, x;
  decode(i, s, z, y, x);
  const int len = axis == 0 ? s.nz : axis == 1 ? s.ny : s.nx;
  const int l = axis == 0 ? z : axis == 1 ? y : x;
  double t = (double)in[i] * w[radius];
  if (l >= radius && l + radius < len) {
    // interior: every tap is in range, constant stride (same sums, same order)
    const size_t stride = axis == 2 ? 1 : axis == 1 ? (size_t)s.nx
                                                    : (size_t)s.ny * s.nx;
    const float* lo = in + i - (size_t)radius * stride;
    const float* hi = in + i + (size_t)radius * stride;
    for (int k = 0; k < radius; ++k) {
      const double pair = (double)lo[(size_t)k * stride] +
                          (double)hi[-(ptrdiff_t)((size_t)k * stride)];
      const double prod = pair * w[k];
      t = t + prod;
    }
  } else {
    for (int ii = -radius; ii < 0; ++ii) {
      const double a = (double)in[neighbour(s, axis, z, y, x,
                                            reflect_index(l + ii, len))];
      const double b = (double)in[neighbour(s, axis, z, y, x,
                                            reflect_index(l - ii, len))];
      const double pair = a + b;
      const double prod = pair * w[ii + radius];
      t = t + prod;
    }
  }
  out[i] = (float)t;
}

__global__ __launch_bounds__(kThreads) void filt_kernel(
    const float* __restrict__ edges, const float* __restrict__ thresh,
    const uint8_t* __restrict__ force_edge, uint8_t* __restrict__ filt,
    size_t n, int* any_non_edge) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  bool f = true;
  if (i < n) {
    f = edges[i] > thresh[i] || (force_edge && force_edge[i]);
    filt[i] = f ? 1 : 0;
  }
  if (__ballot(!f) && (threadIdx.x & 63) == 0) *any_non_edge = 1;
}

// x pass of the EDT: per row, distance (in voxels) to the nearest edge voxel.
__global__ __launch_bounds__(kThreads) void edt_x_kernel(
    const uint8_t* __restrict__ filt, double* __restrict__ d2, Shape s,
    size_t rows, double wx) {
  const size_t r = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (r >= rows) return;
  const uint8_t* f = filt + r * s.nx;
  double* d = d2 + r * s.nx;
  const int kNone = 1 << 30;
  int last = -1;
  for (int x = 0; x < s.nx; ++x) {
    if (f[x]) last = x;
    d[x] = last < 0 ? (double)kNone : (double)(x - last);
  }
  int next = -1;
  for (int x = s.nx - 1; x >= 0; --x) {
    if (f[x]) next = x;
    const int gb = next < 0 ? kNone : next - x;
    const int gf = (int)d[x];
    const int g = gf < gb ? gf : gb;
    if (g >= kNone) {
      d[x] = INFINITY;
    } else {
      const double t = wx * (double)g;
      d[x] = t * t;
    }
  }
}

// y / z pass: lower envelope of parabolas (Felzenszwalb & Huttenlocher) per
// line, one thread per line, stack arrays interleaved across lines.
__global__ __launch_bounds__(kThreads) void edt_line_kernel(
    const double* __restrict__ din, double* __restrict__ dout, Shape s,
    int axis, double w, int* __restrict__ vstack, double* __restrict__ zstack,
    size_t lines) {
  const size_t t = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= lines) return;
  size_t base, stride;
  int len;
  if (axis == 1) {
    const size_t z = t / s.nx, x = t % s.nx;
    base = z * s.ny * s.nx + x;
    stride = s.nx;
    len = s.ny;
  } else {
    base = t;
    stride = (size_t)s.ny * s.nx;
    len = s.nz;
  }
  const double w2 = w * w;
  int k = -1;
  for (int q = 0; q < len; ++q) {
    const double fq = din[base + q * stride];
    if (!(fq < INFINITY)) continue;
    const double qq = (double)q * (double)q;
    const double hq = fq + w2 * qq;
    double sx = -INFINITY;
    while (k >= 0) {
      const int vk = vstack[(size_t)k * lines + t];
      const double fv = din[base + vk * stride];
      const double vv = (double)vk * (double)vk;
      const double hv = fv + w2 * vv;
      sx = (hq - hv) / (2.0 * w2 * (double)(q - vk));
      if (sx <= zstack[(size_t)k * lines + t])
        --k;
      else
        break;
    }
    if (k < 0) {
      k = 0;
      sx = -INFINITY;
    } else {
      ++k;
    }
    vstack[(size_t)k * lines + t] = q;
    zstack[(size_t)k * lines + t] = sx;
  }
  if (k < 0) {
    for (int q = 0; q < len; ++q) dout[base + q * stride] = INFINITY;
    return;
  }
  int j = 0;
  int vj = vstack[t];
  double fj = din[base + vj * stride];
  double znext = j < k ? zstack[(size_t)(j + 1) * lines + t] : INFINITY;
  for (int q = 0; q < len; ++q) {
    while (j < k && znext < (double)q) {
      ++j;
      vj = vstack[(size_t)j * lines + t];
      fj = din[base + vj * stride];
      znext = j < k ? zstack[(size_t)(j + 1) * lines + t] : INFINITY;
    }
    const double dq = (double)(q - vj);
    const double wd = w2 * (dq * dq);
    dout[base + q * stride] = wd + fj;
  }
}

// dt = f32(sqrt(d2)); masked / non-finite -> -1; val = dt + noise * 1e-4 (f64)
__global__ __launch_bounds__(kThreads) void value_kernel(
    const double* __restrict__ d2, const uint8_t* __restrict__ exclude_u8,
    const int* __restrict__ exclude_seg, const double* __restrict__ noise,
    float* __restrict__ dt_out, double* __restrict__ val, size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  float dt = (float)__dsqrt_rn(d2[i]);
  const bool masked = (exclude_u8 && exclude_u8[i]) ||
                      (exclude_seg && exclude_seg[i] > 0);
  if (masked || !(fabsf(dt) < INFINITY)) dt = -1.0f;
  dt_out[i] = dt;
  const double nz = noise[i] * 1e-4;
  val[i] = (double)dt + nz;
}

// 7-wide maximum along one axis, outside = 0.0 (mode='constant', cval=0)
__global__ __launch_bounds__(kThreads) void max7_kernel(
    const double* __restrict__ in, double* __restrict__ out, Shape s, size_t n,
    int axis, int half) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int z, y, x;
  decode(i, s, z, y, x);
  const int len = axis == 0 ? s.nz : axis == 1 ? s.ny : s.nx;
  const int l = axis == 0 ? z : axis == 1 ? y : x;
  double m = in[i];
  for (int o = -half; o <= half; ++o) {
    const int p = l + o;
    const double v = (p < 0 || p >= len)
                         ? 0.0
                         : in[neighbour(s, axis, z, y, x, p)];
    m = v > m ? v : m;
  }
  out[i] = m;
}

// features of the EDT = voxels where the mask is 0
// uint8 canvas image -> its normalised f32 form (table of 256 values made by
// ffn_canvas_create_u8)
__global__ __launch_bounds__(kThreads) void lut_u8_kernel(
    const unsigned char* __restrict__ in, const float* __restrict__ lut,
    float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)kThreads + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kThreads)
    out[i] = lut[in[i]];
}

__global__ __launch_bounds__(kThreads) void mask_to_features_kernel(
    const uint8_t* __restrict__ mask, uint8_t* __restrict__ feat, size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) feat[i] = mask[i] ? 0 : 1;
}

__global__ __launch_bounds__(kThreads) void sqrt_kernel(double* __restrict__ d,
                                                        size_t n) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) d[i] = __dsqrt_rn(d[i]);
}

__global__ __launch_bounds__(kThreads) void peaks_kernel(
    const double* __restrict__ val, const double* __restrict__ mx, Shape s,
    size_t n, int border, int32_t* coords, unsigned cap, unsigned* count) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double v = val[i];
  if (!(v > 0.0) || v != mx[i]) return;
  int z, y, x;
  decode(i, s, z, y, x);
  const int hz = s.nz - border > border ? s.nz - border : border;
  const int hy = s.ny - border > border ? s.ny - border : border;
  const int hx = s.nx - border > border ? s.nx - border : border;
  if (z < border || y < border || x < border || z >= hz || y >= hy || x >= hx)
    return;
  const unsigned k = atomicAdd(count, 1u);
  if (k < cap) {
    coords[3 * k + 0] = z;
    coords[3 * k + 1] = y;
    coords[3 * k + 2] = x;
  }
}

struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct ffn_seeder {
  int device_id = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Buf image, fa, fb, edges, thresh, dt, filt, mask, force, d2a, d2b, val, ma,
      mb, vstack, zstack, coords, small, weights, noise, canvas_f32;
  size_t noise_n = 0;
  int radius = -1;
  size_t last_n = 0;
  double last_ms = 0.0;
};

namespace {

#define S_TRY(expr)                                                           \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess)                                                     \
      return ffn_set_error(FFN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,       \
                           hipGetErrorString(_e), __FILE__, __LINE__);        \
  } while (0)
#define S_OK(expr)                 \
  do {                             \
    int _rc = (expr);              \
    if (_rc != FFN_OK) return _rc; \
  } while (0)

int ensure(Buf& b, size_t bytes) {
  if (b.p && b.bytes >= bytes) return FFN_OK;
  if (b.p) S_TRY(hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  S_TRY(hipMalloc(&b.p, bytes ? bytes : 16));
  b.bytes = bytes ? bytes : 16;
  return FFN_OK;
}

inline dim3 grid_for(size_t n) {
  return dim3((unsigned)((n + kThreads - 1) / kThreads));
}

// The whole pipeline on device pointers.  exclude_u8 / exclude_seg: at most one.
int run_peaks(ffn_seeder* s, const float* d_image, const uint8_t* exclude_u8,
              const int* exclude_seg, const uint8_t* force_edge,
              const int64_t shape_zyx[3], const double voxel[3], size_t cap,
              int32_t* coords_zyx, size_t* n_peaks, int32_t* all_edges) {
  const Shape sh{(int)shape_zyx[0], (int)shape_zyx[1], (int)shape_zyx[2]};
  const size_t n = (size_t)sh.nz * sh.ny * sh.nx;
  if (s->radius < 0)
    return ffn_set_error(FFN_ERR_STATE, "ffn_seeder_set_gaussian not called");
  if (s->noise_n < n)
    return ffn_set_error(FFN_ERR_STATE,
                         "noise holds %zu values, volume needs %zu",
                         s->noise_n, n);
  for (Buf* b : {&s->fa, &s->fb, &s->edges, &s->thresh, &s->dt})
    S_OK(ensure(*b, n * sizeof(float)));
  S_OK(ensure(s->filt, n));
  for (Buf* b : {&s->d2a, &s->d2b, &s->val, &s->ma, &s->mb, &s->zstack})
    S_OK(ensure(*b, n * sizeof(double)));
  S_OK(ensure(s->vstack, n * sizeof(int)));
  S_OK(ensure(s->coords, std::max<size_t>(cap, 1) * 3 * sizeof(int32_t)));
  S_OK(ensure(s->small, 64));
  float* fa = static_cast<float*>(s->fa.p);
  float* fb = static_cast<float*>(s->fb.p);
  float* edges = static_cast<float*>(s->edges.p);
  float* thresh = static_cast<float*>(s->thresh.p);
  uint8_t* filt = static_cast<uint8_t*>(s->filt.p);
  double* d2a = static_cast<double*>(s->d2a.p);
  double* d2b = static_cast<double*>(s->d2b.p);
  double* val = static_cast<double*>(s->val.p);
  double* ma = static_cast<double*>(s->ma.p);
  double* mb = static_cast<double*>(s->mb.p);
  const double* w = static_cast<const double*>(s->weights.p);
  int* any_non_edge = static_cast<int*>(s->small.p);
  unsigned* count = reinterpret_cast<unsigned*>(s->small.p) + 1;
  hipStream_t st = s->stream;
  const dim3 g = grid_for(n), b(kThreads);

  S_TRY(hipMemsetAsync(s->small.p, 0, 64, st));
  S_TRY(hipEventRecord(s->ev0, st));
  // -- Sobel gradient magnitude (generic_gradient_magnitude + sobel) ----------
  for (int axis = 0; axis < 3; ++axis) {
    hipLaunchKernelGGL(tap3_kernel, g, b, 0, st, d_image, fa, sh, n, axis, 0.0,
                       -1.0, 1);
    float* src = fa;
    float* dst = fb;
    for (int other = 0; other < 3; ++other) {
      if (other == axis) continue;
      hipLaunchKernelGGL(tap3_kernel, g, b, 0, st, (const float*)src, dst, sh,
                         n, other, 2.0, 1.0, 0);
      std::swap(src, dst);
    }
    hipLaunchKernelGGL(square_acc_kernel, g, b, 0, st, (const float*)src,
                       edges, n, axis == 0 ? 1 : 0, axis == 2 ? 1 : 0);
  }
  // -- adaptive threshold: gaussian_filter(edges, 49/6), three 1-d passes -----
  hipLaunchKernelGGL(gauss_kernel, g, b, 0, st, (const float*)edges, fa, sh, n,
                     0, w, s->radius);
  hipLaunchKernelGGL(gauss_kernel, g, b, 0, st, (const float*)fa, fb, sh, n, 1,
                     w, s->radius);
  hipLaunchKernelGGL(gauss_kernel, g, b, 0, st, (const float*)fb, thresh, sh,
                     n, 2, w, s->radius);
  hipLaunchKernelGGL(filt_kernel, g, b, 0, st, (const float*)edges,
                     (const float*)thresh, force_edge, filt, n, any_non_edge);
  // -- exact Euclidean distance to the nearest edge voxel ----------------------
  const size_t rows = (size_t)sh.nz * sh.ny;
  hipLaunchKernelGGL(edt_x_kernel, grid_for(rows), b, 0, st,
                     (const uint8_t*)filt, d2a, sh, rows, voxel[2]);
  const size_t ylines = (size_t)sh.nz * sh.nx;
  hipLaunchKernelGGL(edt_line_kernel, grid_for(ylines), b, 0, st,
                     (const double*)d2a, d2b, sh, 1, voxel[1],
                     static_cast<int*>(s->vstack.p),
                     static_cast<double*>(s->zstack.p), ylines);
  const size_t zlines = (size_t)sh.ny * sh.nx;
  hipLaunchKernelGGL(edt_line_kernel, grid_for(zlines), b, 0, st,
                     (const double*)d2b, d2a, sh, 0, voxel[0],
                     static_cast<int*>(s->vstack.p),
                     static_cast<double*>(s->zstack.p), zlines);
  // -- peaks ---------------------------------------------------------------------
  hipLaunchKernelGGL(value_kernel, g, b, 0, st, (const double*)d2a, exclude_u8,
                     exclude_seg, static_cast<const double*>(s->noise.p),
                     static_cast<float*>(s->dt.p), val, n);
  hipLaunchKernelGGL(max7_kernel, g, b, 0, st, (const double*)val, ma, sh, n, 2,
                     3);
  hipLaunchKernelGGL(max7_kernel, g, b, 0, st, (const double*)ma, mb, sh, n, 1,
                     3);
  hipLaunchKernelGGL(max7_kernel, g, b, 0, st, (const double*)mb, ma, sh, n, 0,
                     3);
  hipLaunchKernelGGL(peaks_kernel, g, b, 0, st, (const double*)val,
                     (const double*)ma, sh, n, 3,
                     static_cast<int32_t*>(s->coords.p), (unsigned)cap, count);
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(s->ev1, st));
  int host[2] = {0, 0};
  S_TRY(hipMemcpyAsync(host, s->small.p, 8, hipMemcpyDeviceToHost, st));
  S_TRY(hipStreamSynchronize(st));
  float ms = 0.f;
  S_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  s->last_ms = ms;
  s->last_n = n;
  if (all_edges) *all_edges = host[0] ? 0 : 1;
  if (!host[0]) {  // every voxel is an edge: no seeds (seed.py:178-179)
    *n_peaks = 0;
    return FFN_OK;
  }
  const unsigned found = (unsigned)host[1];
  *n_peaks = found;
  if (found > cap)
    return ffn_set_error(FFN_ERR_ARG, "%u peaks exceed cap %zu", found, cap);
  if (found)
    S_TRY(hipMemcpy(coords_zyx, s->coords.p, (size_t)found * 3 * sizeof(int32_t),
                    hipMemcpyDeviceToHost));
  return FFN_OK;
}

int check_shape(const int64_t shape_zyx[3], size_t* n) {
  if (!shape_zyx) return ffn_set_error(FFN_ERR_ARG, "NULL shape");
  double nd = 1.0;
  for (int k = 0; k < 3; ++k) {
    if (shape_zyx[k] <= 0)
      return ffn_set_error(FFN_ERR_ARG, "shape must be positive");
    nd *= (double)shape_zyx[k];
  }
  if (nd >= 2147483648.0)
    return ffn_set_error(FFN_ERR_ARG, "volume must have < 2^31 voxels");
  *n = (size_t)nd;
  return FFN_OK;
}

}  // namespace

extern "C" {

int ffn_seeder_create(int device_id, ffn_seeder** out) {
  if (!out) return ffn_set_error(FFN_ERR_ARG, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  S_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev)
    return ffn_set_error(FFN_ERR_ARG, "device %d not present (%d devices)",
                         device_id, ndev);
  S_TRY(hipSetDevice(device_id));
  ffn_seeder* s = new ffn_seeder();
  s->device_id = device_id;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&s->ev0);
  if (e == hipSuccess) e = hipEventCreate(&s->ev1);
  if (e != hipSuccess) {
    ffn_seeder_destroy(s);
    return ffn_set_error(FFN_ERR_HIP, "stream/event creation failed: %s",
                         hipGetErrorString(e));
  }
  *out = s;
  return FFN_OK;
}

void ffn_seeder_destroy(ffn_seeder* s) {
  if (!s) return;
  (void)hipSetDevice(s->device_id);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (Buf* b : {&s->image, &s->fa, &s->fb, &s->edges, &s->thresh, &s->dt,
                 &s->filt, &s->mask, &s->force, &s->d2a, &s->d2b, &s->val,
                 &s->ma, &s->mb, &s->vstack, &s->zstack, &s->coords, &s->small,
                 &s->weights, &s->noise, &s->canvas_f32})
    if (b->p) (void)hipFree(b->p);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int ffn_seeder_set_noise(ffn_seeder* s, const double* noise, size_t n) {
  if (!s || (n && !noise)) return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  S_TRY(hipSetDevice(s->device_id));
  S_OK(ensure(s->noise, n * sizeof(double)));
  if (n)
    S_TRY(hipMemcpy(s->noise.p, noise, n * sizeof(double),
                    hipMemcpyHostToDevice));
  s->noise_n = n;
  return FFN_OK;
}

int ffn_seeder_set_gaussian(ffn_seeder* s, const double* weights, int radius) {
  if (!s || !weights) return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (radius < 0 || radius > kMaxRadius)
    return ffn_set_error(FFN_ERR_ARG, "radius must be in [0, %d]", kMaxRadius);
  S_TRY(hipSetDevice(s->device_id));
  const size_t bytes = (size_t)(2 * radius + 1) * sizeof(double);
  S_OK(ensure(s->weights, bytes));
  S_TRY(hipMemcpy(s->weights.p, weights, bytes, hipMemcpyHostToDevice));
  s->radius = radius;
  return FFN_OK;
}

int ffn_seeder_peaks(ffn_seeder* s, const float* image, const uint8_t* exclude,
                     const uint8_t* force_edge, const int64_t shape_zyx[3],
                     const double voxel_size_zyx[3], size_t cap,
                     int32_t* coords_zyx, size_t* n_peaks, int32_t* all_edges) {
  if (!s || !image || !voxel_size_zyx || !n_peaks || (cap && !coords_zyx))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  size_t n = 0;
  S_OK(check_shape(shape_zyx, &n));
  S_TRY(hipSetDevice(s->device_id));
  S_OK(ensure(s->image, n * sizeof(float)));
  S_TRY(hipMemcpyAsync(s->image.p, image, n * sizeof(float),
                       hipMemcpyHostToDevice, s->stream));
  if (exclude) {
    S_OK(ensure(s->mask, n));
    S_TRY(hipMemcpyAsync(s->mask.p, exclude, n, hipMemcpyHostToDevice,
                         s->stream));
  }
  if (force_edge) {
    S_OK(ensure(s->force, n));
    S_TRY(hipMemcpyAsync(s->force.p, force_edge, n, hipMemcpyHostToDevice,
                         s->stream));
  }
  return run_peaks(s, static_cast<const float*>(s->image.p),
                   exclude ? static_cast<const uint8_t*>(s->mask.p) : nullptr,
                   nullptr,
                   force_edge ? static_cast<const uint8_t*>(s->force.p)
                              : nullptr,
                   shape_zyx, voxel_size_zyx, cap, coords_zyx, n_peaks,
                   all_edges);
}

int ffn_seeder_peaks_canvas(ffn_seeder* s, ffn_canvas* canvas,
                            const double voxel_size_zyx[3], size_t cap,
                            int32_t* coords_zyx, size_t* n_peaks,
                            int32_t* all_edges) {
  if (!s || !canvas || !voxel_size_zyx || !n_peaks || (cap && !coords_zyx))
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  FfnCanvasView v;
  S_OK(ffn_canvas_view(canvas, &v));
  if (v.device_id != s->device_id)
    return ffn_set_error(FFN_ERR_ARG, "canvas lives on device %d, seeder on %d",
                         v.device_id, s->device_id);
  S_TRY(hipSetDevice(s->device_id));
  // the canvas' own stream may still be pasting / committing
  S_TRY(hipStreamSynchronize(static_cast<hipStream_t>(v.engine_stream)));
  const int64_t shape[3] = {v.shape_zyx[0], v.shape_zyx[1], v.shape_zyx[2]};
  size_t n = 0;
  S_OK(check_shape(shape, &n));
  const float* image = v.image;
  if (!image) {
    // uint8 canvas: PolicyPeaks works on the normalised f32 image (seed.py:146);
    // materialise it for the duration of this call only
    S_OK(ensure(s->canvas_f32, n * sizeof(float)));
    hipLaunchKernelGGL(lut_u8_kernel, grid_for(n), dim3(kThreads), 0, s->stream,
                       v.image_u8, v.image_lut,
                       static_cast<float*>(s->canvas_f32.p), n);
    image = static_cast<const float*>(s->canvas_f32.p);
  }
  return run_peaks(s, image, nullptr, v.segmentation, nullptr, shape,
                   voxel_size_zyx, cap, coords_zyx, n_peaks, all_edges);
}

int ffn_seeder_edt(ffn_seeder* s, const uint8_t* mask,
                   const int64_t shape_zyx[3], const double voxel_size_zyx[3],
                   double* dist) {
  if (!s || !mask || !voxel_size_zyx || !dist)
    return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  size_t n = 0;
  S_OK(check_shape(shape_zyx, &n));
  S_TRY(hipSetDevice(s->device_id));
  const Shape sh{(int)shape_zyx[0], (int)shape_zyx[1], (int)shape_zyx[2]};
  S_OK(ensure(s->mask, n));
  S_OK(ensure(s->filt, n));
  S_OK(ensure(s->d2a, n * sizeof(double)));
  S_OK(ensure(s->d2b, n * sizeof(double)));
  S_OK(ensure(s->zstack, n * sizeof(double)));
  S_OK(ensure(s->vstack, n * sizeof(int)));
  hipStream_t st = s->stream;
  const dim3 b(kThreads);
  S_TRY(hipMemcpyAsync(s->mask.p, mask, n, hipMemcpyHostToDevice, st));
  S_TRY(hipEventRecord(s->ev0, st));
  hipLaunchKernelGGL(mask_to_features_kernel, grid_for(n), b, 0, st,
                     static_cast<const uint8_t*>(s->mask.p),
                     static_cast<uint8_t*>(s->filt.p), n);
  double* d2a = static_cast<double*>(s->d2a.p);
  double* d2b = static_cast<double*>(s->d2b.p);
  const size_t rows = (size_t)sh.nz * sh.ny;
  hipLaunchKernelGGL(edt_x_kernel, grid_for(rows), b, 0, st,
                     static_cast<const uint8_t*>(s->filt.p), d2a, sh, rows,
                     voxel_size_zyx[2]);
  const size_t ylines = (size_t)sh.nz * sh.nx;
  hipLaunchKernelGGL(edt_line_kernel, grid_for(ylines), b, 0, st,
                     (const double*)d2a, d2b, sh, 1, voxel_size_zyx[1],
                     static_cast<int*>(s->vstack.p),
                     static_cast<double*>(s->zstack.p), ylines);
  const size_t zlines = (size_t)sh.ny * sh.nx;
  hipLaunchKernelGGL(edt_line_kernel, grid_for(zlines), b, 0, st,
                     (const double*)d2b, d2a, sh, 0, voxel_size_zyx[0],
                     static_cast<int*>(s->vstack.p),
                     static_cast<double*>(s->zstack.p), zlines);
  hipLaunchKernelGGL(sqrt_kernel, grid_for(n), b, 0, st, d2a, n);
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(s->ev1, st));
  S_TRY(hipMemcpyAsync(dist, d2a, n * sizeof(double), hipMemcpyDeviceToHost, st));
  S_TRY(hipStreamSynchronize(st));
  float ms = 0.f;
  S_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  s->last_ms = ms;
  s->last_n = 0;  // no peaks stages to read back
  return FFN_OK;
}

int ffn_seeder_read_stage(ffn_seeder* s, int which, float* dst) {
  if (!s || !dst) return ffn_set_error(FFN_ERR_ARG, "NULL argument");
  if (!s->last_n) return ffn_set_error(FFN_ERR_STATE, "no peaks call yet");
  const Buf* b = which == 0 ? &s->edges : which == 1 ? &s->thresh
                 : which == 2 ? &s->dt : nullptr;
  if (!b) return ffn_set_error(FFN_ERR_ARG, "which must be 0, 1 or 2");
  S_TRY(hipSetDevice(s->device_id));
  S_TRY(hipMemcpy(dst, b->p, s->last_n * sizeof(float), hipMemcpyDeviceToHost));
  return FFN_OK;
}

int ffn_seeder_last_timing(ffn_seeder* s, double* kernel_ms, double* voxels) {
  if (!s) return ffn_set_error(FFN_ERR_ARG, "NULL handle");
  if (kernel_ms) *kernel_ms = s->last_ms;
  if (voxels) *voxels = (double)s->last_n;
  return FFN_OK;
}

}  // extern "C"
