// Device kernels of the MI355X (gfx950 / CDNA4) FFN field-of-view engine.
//
// Written for gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x4_f32 (exact f32
// MFMA, bitwise an fmaf chain), 160 KiB LDS per CU, 256 CUs in 8 XCDs.
//
// Activation layout in HBM ("padded flat", channels last):
//   position p = z*plane + y*XS + x,  XS = fx+1, plane = (fy+1)*XS
//   one zero column (x = fx) and one zero row (y = fy) are shared between
//   neighbouring rows / planes, and a zero guard of plane+XS+1 positions sits
//   in front of and behind the FoV, so EVERY 3x3x3 tap of EVERY position is a
//   plain constant offset  dz*plane + dy*XS + dx  -- the SAME zero padding of
//   tf_slim.convolution3d (reference convstack_3d.py:28-31) without a single
//   bounds test in the inner loop.  Invalid (padding) positions are never
//   written, so they stay zero for the lifetime of the engine.
//   Each position holds 32 channels = 128 B = one cache line.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ffn_hip.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFeatures = 32;
constexpr int kChunk = 160;          // output positions per workgroup
constexpr int kTile = 16;            // positions per MFMA M-tile
constexpr int kTilesPerWave = 5;     // 2 tile groups x 5 tiles = 10 tiles = kChunk
constexpr int kConvThreads = 256;    // 4 waves: (nhalf, tile group)

struct Geom {
  int fz, fy, fx;      // FoV (zyx)
  int dz, dy, dx;      // deltas (zyx)
  int XS, plane;       // padded strides (positions)
  int npos;            // fz * plane
  int guard;           // plane + XS + 1
  int nchunks;         // ceil(npos / kChunk)
  int V;               // fz*fy*fx
  int R;               // LDS rows per dz segment = kChunk + 2*(XS+1)
  long act_stride;     // floats per FoV activation buffer
  // The split-product kernels may lay the FoV out with its axes permuted (the
  // shortest one as the row direction): axis a of THIS geometry is axis oa[a]
  // (0 z, 1 y, 2 x) of the caller's FoV / of the canvas, and one step along it
  // moves dstr[a] voxels in the caller's dense [z][y][x] order.  Identity:
  // oa = {0, 1, 2}, dstr = {fy fx, fx, 1}.
  int oa[3];
  int dstr[3];
  // A model that predicts a SMALLER mask than the seed it reads (ModelInfo
  // pred_mask_size < input_seed_size, reference model.py:168-183, inference.py:
  // 218,410-411): the canvas step scores, counts and pastes only the centred
  // box [c0, c1) of the FoV (caller's zyx); crop = 0: the whole FoV.
  int crop;
  int c0[3], c1[3];
  int Vp;              // voxels of the box (= V without a crop)
};

__device__ __forceinline__ bool in_pred_box(const Geom& g, int z, int y, int x) {
  return z >= g.c0[0] && z < g.c1[0] && y >= g.c0[1] && y < g.c1[1] && x >= g.c0[2] &&
         x < g.c1[2];
}

// Per-FoV step descriptor read by the gather / paste kernels.
struct StepItem {
  const float* image;        // f32 canvas image (already normalised), or NULL:
  const uint8_t* image_u8;   //   raw uint8 image ...
  const float* image_lut;    //   ... and its 256-entry normalisation table
  float* seed;
  const int32_t* seg;
  int cz, cy, cx;
  ffn_step_request req;
};

// ---------------------------------------------------------------------------
// conv0a: gather + concat(image, seed) -> 3x3x3 conv 2->32 + bias + ReLU
// (reference inference.py:348-354,399-407; convstack_3d.py:38,86).
//
// Reads the FoV straight out of the canvas volumes (or, for the stateless
// predict path, out of the uploaded dense FoV treated as a FoV-sized canvas),
// substitutes pad_value for NaN ("never visited") seed voxels, and also writes
// the raw (NaN-preserving) seed FoV to `seed_raw`, which the head (seed + update)
// and the paste kernel (disco mask) need later.  K = 54 only: VALU.
// One block = a 4x8x8 tile of positions: the tile + halo is staged once in LDS
// (one canvas read per input voxel), every thread then computes all 32 output
// channels of its position with the weights coming through the scalar cache.
// ---------------------------------------------------------------------------
struct StepItems {
  const StepItem* items;  // device array (batched path)
  StepItem inline_item;   // kernarg copy (single-canvas fast path)
  int use_inline;
};

// What the gather / faces / paste kernels read of a StepItem, in registers: with
// `const StepItem& it = inline ? kernarg copy : items[item]` every field access
// is a flat load behind a select (a dependent memory round trip each); here the
// single-FoV path reads its fields straight from the kernel arguments.
// (global address space spelled out: through generic pointers these would be
// flat loads, which the compiler orders against every LDS access)
#define FFN_GLOBAL __attribute__((address_space(1)))
struct ItemView {
  const FFN_GLOBAL float* image;
  const FFN_GLOBAL uint8_t* image_u8;
  const FFN_GLOBAL float* image_lut;
  FFN_GLOBAL float* seed;
  const FFN_GLOBAL int32_t* seg;
  int cz, cy, cx;
  int pos[3];
  const ffn_step_request* req;  // start_pos / candidates (read per lane)
};
__device__ __forceinline__ ItemView item_view(const StepItems& si, int item) {
  ItemView v;
#define FFN_VIEW_FROM(S)                                                        \
  v.image = (const FFN_GLOBAL float*)(S).image;                                  \
  v.image_u8 = (const FFN_GLOBAL uint8_t*)(S).image_u8;                          \
  v.image_lut = (const FFN_GLOBAL float*)(S).image_lut;                          \
  v.seed = (FFN_GLOBAL float*)(S).seed;                                          \
  v.seg = (const FFN_GLOBAL int32_t*)(S).seg;                                    \
  v.cz = (S).cz, v.cy = (S).cy, v.cx = (S).cx;                                   \
  v.pos[0] = (S).req.pos[0], v.pos[1] = (S).req.pos[1], v.pos[2] = (S).req.pos[2]; \
  v.req = &(S).req;
  // (the position is pinned on its side of the select, so that it is read from
  // the kernel arguments there and not through the merged `req` pointer)
  if (si.use_inline) {
    FFN_VIEW_FROM(si.inline_item)
    asm volatile("" : "+s"(v.pos[0]), "+s"(v.pos[1]), "+s"(v.pos[2]));
  } else {
    FFN_VIEW_FROM(si.items[item])
    asm volatile("" : "+v"(v.pos[0]), "+v"(v.pos[1]), "+v"(v.pos[2]));
  }
#undef FFN_VIEW_FROM
  return v;
}

constexpr int kC0Z = 4, kC0Y = 8, kC0X = 8;  // conv0a output tile per block
constexpr int kC0Threads = 512;               // 256 positions x 2 cout halves

// conv0_a on the matrix cores: the 4 x 8 x 8 output tile + halo is staged once
// in LDS (one canvas read per input voxel; gfx.oa / canvas strides map this
// layout's axes onto the canvas'), then an implicit GEMM with K = 27 taps x 2 channels = 54 (padded to 56 = 14
// k-steps of v_mfma_f32_16x16x4_f32).  A block = 256 positions = 16 M-tiles;
// wave w owns M-tiles 2w, 2w+1 for both cout halves (56 MFMAs).  A operand: one
// ds_read_b32 per k-step straight from the (image, seed) tile (lane group g
// reads channel g & 1 of tap 2s + (g >> 1)); B operand: the [54][32] weights,
// 28 registers per lane, loaded once.  5x fewer issue cycles than the VALU form.
// SPLIT (conv_variant 6): the output leaves as "split planes" (fp16 hi + scaled
// residual, 16 B per position and chunk plane; see conv32d) instead of f32.
struct Conv0SplitOut {
  char* out_sp;            // position 0 of plane 0, item 0
  long sp_plane_bytes;     // positions x 16
  long item_bytes;
  unsigned* range_flag;
  unsigned range_tag;
};
typedef _Float16 f16x8_c0 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_c0 __attribute__((ext_vector_type(4)));

// Speculative launch (single-FoV steps of the library's segment loop): conv0_a of
// the NEXT step is queued behind this step's paste, before the host has seen this
// step's result, for the first of up to kSpecMax queued positions that passes
// Canvas.is_valid_pos's device part (inference.py:325,341: NOT seed < move
// threshold, segmentation <= 0; the bounds part is the host's, done before the
// launch).  The host makes the same choice from the same values one turn-around
// later (ffn_step_result.cand_seed / cand_seg) and then queues the rest of the
// step behind this launch; `choice` (-1: none valid, nothing computed) lets the
// step's faces kernel verify that both chose the same position.
constexpr int kSpecMax = 3;
struct SpecArgs {
  int n;                 // 0: a normal launch at si's request position
  int pos[kSpecMax][3];  // zyx
  float move_thr;
  int* choice;
};

// The canvas as it WILL be once the step whose paste runs next to this conv0_a
// (the fused faces + paste + next conv0_a launch) has pasted: inside that step's
// prediction box the seed is post_disco(logits, old seed) -- exactly what its
// paste blocks are writing meanwhile -- elsewhere the canvas itself.  on = 0: the
// canvas as it is (a launch of its own, or a void step that pastes nothing).
struct SeedOverlay {
  int on;
  int disco;
  const float* lg;   // the step's logits, dense [z][y][x] of the caller's FoV
  const float* old;  // its raw input seed
  int z0, y0, x0;    // canvas corner of its FoV
  int fy, fx;        // its FoV's row / plane strides
  int c0[3], c1[3];  // its prediction box (Geom::c0 / c1)
};

__device__ __forceinline__ float post_disco(float lg, float old, bool disco);

// index into the overlay's dense arrays of canvas voxel (Z, Y, X), or -1
__device__ __forceinline__ int overlay_index(const SeedOverlay& ov, int Z, int Y, int X) {
  const int lz = Z - ov.z0, ly = Y - ov.y0, lx = X - ov.x0;
  const bool in = ov.on && lz >= ov.c0[0] && lz < ov.c1[0] && ly >= ov.c0[1] &&
                  ly < ov.c1[1] && lx >= ov.c0[2] && lx < ov.c1[2];
  return in ? (lz * ov.fy + ly) * ov.fx + lx : -1;
}

template <bool SPLIT>
__device__ __forceinline__ void conv0a_body(
    const int tile_block, const int item, const StepItems& si, float pad_value,
    const float* __restrict__ w /*[27][2][32]*/,
    const float* __restrict__ bias, float* __restrict__ out,
    float* __restrict__ seed_raw, const Geom& g, int tiles_y, int tiles_x,
    const Conv0SplitOut& so, const SpecArgs& sp, const SeedOverlay& ov) {
  constexpr int HZ = kC0Z + 2, HY = kC0Y + 2, HX = kC0X + 2;
  __shared__ float tile[HZ * HY * HX * 2];  // (image, seed) interleaved
  __shared__ float s_lut[256];              // uint8 canvases: normalisation table
  // SPLIT: the block's 256 x 32 outputs, transposed through LDS (36-float rows)
  __shared__ __attribute__((aligned(16))) float otile[SPLIT ? 256 * 36 : 4];
  const ItemView it = item_view(si, item);
  int b = tile_block;
  const int tx = b % tiles_x;
  b /= tiles_x;
  const int ty = b % tiles_y;
  const int tz = b / tiles_y;
  const int oz = tz * kC0Z, oy = ty * kC0Y, ox = tx * kC0X;  // FoV coords
  // canvas strides of this geometry's axes (axis a = canvas axis g.oa[a])
  const long cstr[3] = {(long)it.cy * it.cx, (long)it.cx, 1};
  const long sz = cstr[g.oa[0]], sy = cstr[g.oa[1]], sx = cstr[g.oa[2]];
  int pos[3] = {it.pos[0], it.pos[1], it.pos[2]};

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15;   // A row (position) / B column (cout) of this lane
  const int grp = lane >> 4;  // k index inside a k-step

  // Gather: every canvas load of the block is issued before any is waited for
  // (<= kC0Per elements per thread), and a uint8 canvas' normalisation table
  // ((x - mean) / stddev of runner.py:383-385 as a 256-entry look-up) goes
  // through LDS -- one global round trip for the whole gather instead of one
  // per pass and another per look-up.
  constexpr int kC0Per = (HZ * HY * HX + kC0Threads - 1) / kC0Threads;
  const bool u8 = it.image == nullptr;
  float g_img[kC0Per] = {}, g_seed[kC0Per];
  unsigned g_raw[kC0Per] = {};
  float o_l[kC0Per] = {}, o_o[kC0Per] = {};
  int g_ov[kC0Per];  // index into the overlay (a voxel the running paste writes), or -1
  long g_out[kC0Per];  // seed_raw index of an interior voxel, else -1
  bool g_in[kC0Per];
  int g_zz[kC0Per], g_yy[kC0Per], g_xx[kC0Per];
#pragma unroll
  for (int k = 0; k < kC0Per; ++k) {
    const int e = threadIdx.x + k * kC0Threads;
    const int ec = e < HZ * HY * HX ? e : 0;
    const int hx = ec % HX;
    const int t = ec / HX;
    const int hy = t % HY;
    const int hz = t / HY;
    const int zz = oz + hz - 1, yy = oy + hy - 1, xx = ox + hx - 1;
    g_zz[k] = zz, g_yy[k] = yy, g_xx[k] = xx;
    g_in[k] = e < HZ * HY * HX && zz >= 0 && zz < g.fz && yy >= 0 && yy < g.fy &&
              xx >= 0 && xx < g.fx;
    // interior voxel: keep the raw seed (NaN preserved), at its place in the
    // caller's dense [z][y][x] order
    g_out[k] = (g_in[k] && hz >= 1 && hz <= kC0Z && hy >= 1 && hy <= kC0Y &&
                hx >= 1 && hx <= kC0X)
                   ? (long)((size_t)item * g.V + (size_t)zz * g.dstr[0] +
                            yy * g.dstr[1] + xx * g.dstr[2])
                   : -1;
  }
  // the loads of the FoV at canvas position p3 (zyx), all in flight at once
  auto issue_gather = [&](const int* p3) {
    const int pz = g.oa[0] == 0 ? p3[0] : g.oa[0] == 1 ? p3[1] : p3[2];
    const int py = g.oa[1] == 0 ? p3[0] : g.oa[1] == 1 ? p3[1] : p3[2];
    const int px = g.oa[2] == 0 ? p3[0] : g.oa[2] == 1 ? p3[1] : p3[2];
    const int z0 = pz - g.fz / 2;
    const int y0 = py - g.fy / 2;
    const int x0 = px - g.fx / 2;
    size_t g_ci[kC0Per];
#pragma unroll
    for (int k = 0; k < kC0Per; ++k) {
      // (voxel 0 of the canvas stands in outside the FoV: loads without branches)
      g_ci[k] = g_in[k] ? (size_t)((z0 + g_zz[k]) * sz + (y0 + g_yy[k]) * sy +
                                   (x0 + g_xx[k]) * sx)
                        : 0;
      g_ov[k] = -1;
      if (ov.on && g_in[k]) {
        int cc[3];  // canvas coordinates: this geometry's axis a is canvas axis oa[a]
        cc[g.oa[0]] = z0 + g_zz[k];
        cc[g.oa[1]] = y0 + g_yy[k];
        cc[g.oa[2]] = x0 + g_xx[k];
        g_ov[k] = overlay_index(ov, cc[0], cc[1], cc[2]);
      }
    }
    if (u8) {
#pragma unroll
      for (int k = 0; k < kC0Per; ++k) g_raw[k] = it.image_u8[g_ci[k]];
    } else {
#pragma unroll
      for (int k = 0; k < kC0Per; ++k) g_img[k] = it.image[g_ci[k]];
    }
#pragma unroll
    for (int k = 0; k < kC0Per; ++k) g_seed[k] = it.seed[g_ci[k]];
    if (ov.on) {
#pragma unroll
      for (int k = 0; k < kC0Per; ++k) {
        o_l[k] = ov.lg[g_ov[k] < 0 ? 0 : g_ov[k]];
        o_o[k] = ov.old[g_ov[k] < 0 ? 0 : g_ov[k]];
      }
    }
  };

  if (sp.n > 0) {  // every block makes the same choice from the same loads
    // (all of them in flight at once)
    float sv[kSpecMax];
    int gv[kSpecMax];
    int ovi[kSpecMax];
    float ol[kSpecMax], oo[kSpecMax];
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) {
      const size_t ci =  // (the host fills unused slots with candidate 0)
          ((size_t)sp.pos[k][0] * it.cy + sp.pos[k][1]) * it.cx + sp.pos[k][2];
      sv[k] = it.seed[ci];
      gv[k] = it.seg[ci];
      ovi[k] = overlay_index(ov, sp.pos[k][0], sp.pos[k][1], sp.pos[k][2]);
      ol[k] = ov.on ? ov.lg[ovi[k] < 0 ? 0 : ovi[k]] : 0.f;
      oo[k] = ov.on ? ov.old[ovi[k] < 0 ? 0 : ovi[k]] : 0.f;
    }
    // ... and behind them, before their values are back, the gather for the FIRST
    // position of the list: it is the one chosen unless the step about to end
    // has invalidated it, and then its round trip is the choice's own
    issue_gather(sp.pos[0]);
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k)
      if (ovi[k] >= 0) sv[k] = post_disco(ol[k], oo[k], ov.disco != 0);
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k)  // (no short-circuit into dependent loads)
      asm volatile("" : "+v"(sv[k]), "+v"(gv[k]));
    int ch = -1;
#pragma unroll
    for (int k = kSpecMax - 1; k >= 0; --k)
      if (k < sp.n && !(sv[k] < sp.move_thr) && gv[k] <= 0) ch = k;
    if (tile_block == 0 && threadIdx.x == 0) *sp.choice = ch;
    if (ch < 0) return;
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k)
      if (k == ch) {
        pos[0] = sp.pos[k][0];
        pos[1] = sp.pos[k][1];
        pos[2] = sp.pos[k][2];
      }
    if (ch != 0) issue_gather(pos);  // (every block and lane alike)
  } else {
    issue_gather(pos);
  }

  // B fragments: k = 4 s + grp -> w[k][16 nhalf + i]; k >= 54 is zero padding
  float bw[2][14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int kk = 4 * s + grp;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      bw[h][s] = kk < 54 ? w[kk * kFeatures + 16 * h + i] : 0.0f;
  }
  const float bias0 = bias[i], bias1 = bias[16 + i];
  float lut_v = 0.0f;  // in flight with the canvas loads
  if (u8 && threadIdx.x < 256) lut_v = it.image_lut[threadIdx.x];
  if (ov.on) {
#pragma unroll
    for (int k = 0; k < kC0Per; ++k)
      if (g_ov[k] >= 0) g_seed[k] = post_disco(o_l[k], o_o[k], ov.disco != 0);
  }
  if (u8 && threadIdx.x < 256) s_lut[threadIdx.x] = lut_v;
  __syncthreads();  // the table is in LDS
#pragma unroll
  for (int k = 0; k < kC0Per; ++k) {
    const int e = threadIdx.x + k * kC0Threads;
    if (e >= HZ * HY * HX) continue;
    float vi = 0.0f, vs = 0.0f;  // SAME zero padding outside the FoV
    if (g_in[k]) {
      vi = u8 ? s_lut[g_raw[k]] : g_img[k];
      vs = g_seed[k];
      if (g_out[k] >= 0) seed_raw[g_out[k]] = vs;
      if (vs != vs) vs = pad_value;  // NaN -> pad (inference.py:406-407)
    }
    tile[2 * e] = vi;
    tile[2 * e + 1] = vs;
  }
  __syncthreads();

  const int ch = grp & 1;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int lp = (wave * 2 + m) * kTile + i;  // this lane's A row
    const int lx = lp % kC0X;
    const int ly = (lp / kC0X) % kC0Y;
    const int lz = lp / (kC0X * kC0Y);
    const int abase = ((lz * HY + ly) * HX + lx) * 2 + ch;
    f32x4 acc0 = {bias0, bias0, bias0, bias0};
    f32x4 acc1 = {bias1, bias1, bias1, bias1};
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      int tap = 2 * s + (grp >> 1);
      tap = tap > 26 ? 26 : tap;  // k = 54, 55: weight is zero, any finite A
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const float av = tile[abase + ((kz * HY + ky) * HX + kx) * 2];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[0][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[1][s], acc1, 0, 0, 0);
    }
    // D fragment: lane (i, grp) holds cout i of positions 4 grp + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int op = (wave * 2 + m) * kTile + grp * 4 + r;
      if constexpr (SPLIT) {
        otile[op * 36 + i] = fmaxf(acc0[r], 0.0f);
        otile[op * 36 + 16 + i] = fmaxf(acc1[r], 0.0f);
        continue;
      }
      const int ox_ = op % kC0X;
      const int oy_ = (op / kC0X) % kC0Y;
      const int oz_ = op / (kC0X * kC0Y);
      const int z = oz + oz_, y = oy + oy_, x = ox + ox_;
      if (z >= g.fz || y >= g.fy || x >= g.fx) continue;
      const size_t p = (size_t)z * g.plane + (size_t)y * g.XS + x;
      float* o = out + (size_t)item * g.act_stride + p * kFeatures;
      o[i] = fmaxf(acc0[r], 0.0f);
      o[16 + i] = fmaxf(acc1[r], 0.0f);
    }
  }
  if constexpr (SPLIT) {
    __syncthreads();
    unsigned range_max = 0;
    char* ob = so.out_sp + (long)item * so.item_bytes;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = threadIdx.x + kC0Threads * k;  // (chunk plane c, position op)
      const int c = e >> 8, op = e & 255;
      const int ox_ = op % kC0X;
      const int oy_ = (op / kC0X) % kC0Y;
      const int oz_ = op / (kC0X * kC0Y);
      const int z = oz + oz_, y = oy + oy_, x = ox + ox_;
      if (z >= g.fz || y >= g.fy || x >= g.fx) continue;
      const long p = (long)z * g.plane + (long)y * g.XS + x;
      const f32x4 va = *reinterpret_cast<const f32x4*>(otile + op * 36 + c * 8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(otile + op * 36 + c * 8 + 4);
      f16x8_c0 hi, res;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 v = h ? vb : va;
        f32x4 vh = v;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const unsigned mbits = __float_as_uint(v[cc]) & 0x7fffffffu;
          range_max = mbits > range_max ? mbits : range_max;
          vh[cc] = mbits < 0x38800000u ? 0.0f : v[cc];  // |x| < 2^-14
        }
        const f16x4_c0 h4 = __builtin_convertvector(vh, f16x4_c0);
        const f32x4 r1 = (v - __builtin_convertvector(h4, f32x4)) * 2048.0f;
        const f16x4_c0 r4 = __builtin_convertvector(r1, f16x4_c0);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          hi[4 * h + cc] = h4[cc];
          res[4 * h + cc] = r4[cc];
        }
      }
      *reinterpret_cast<f16x8_c0*>(ob + (long)c * so.sp_plane_bytes + p * 16) = hi;
      *reinterpret_cast<f16x8_c0*>(ob + (long)(4 + c) * so.sp_plane_bytes + p * 16) =
          res;
    }
    if (__ballot(range_max > 0x477fe000u) && lane == 0)  // > 65504 (or NaN)
      *so.range_flag = so.range_tag;
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(kC0Threads) void conv0a_mfma_kernel(
    StepItems si, float pad_value, const float* __restrict__ w /*[27][2][32]*/,
    const float* __restrict__ bias, float* __restrict__ out,
    float* __restrict__ seed_raw, Geom g, int tiles_y, int tiles_x,
    Conv0SplitOut so, SpecArgs sp) {
  SeedOverlay ov;
  ov.on = 0;
  conv0a_body<SPLIT>(blockIdx.x, blockIdx.y, si, pad_value, w, bias, out, seed_raw, g,
                     tiles_y, tiles_x, so, sp, ov);
}

// ---------------------------------------------------------------------------
// conv32: 3x3x3 conv 32->32 as an implicit GEMM on the exact-f32 MFMA
// (reference convstack_3d.py:39,45-47; 23 of the 24 convs of a depth-12 stack,
// 99.7 % of the FLOPs).
//
//   M = positions (16 per MFMA tile), N = 32 couts (two halves of 16),
//   K = 27 taps x 32 cin  (8 k-steps of 4 per tap).
//
// Workgroup = 4 waves = one chunk of 160 consecutive padded positions.
//   wave w: nhalf = w & 1 (which 16 couts), tile group = w >> 1 (which 5 tiles)
//   -> 5 independent accumulator chains per wave (f32x4 each): the 40-cycle
//      dependent latency of v_mfma_f32_16x16x4_f32 never stalls the 32-cycle
//      issue rate.
// Operands:
//   A (activations): the chunk plus its halo (3 dz-segments of R rows x 128 B)
//      is staged ONCE into LDS (ReLU fused into the staging when RELU_IN); the
//      16-byte quads of a row are XOR-swizzled with (row & 7) so that the
//      ds_read_b128 of 16 consecutive rows is bank-conflict free for every tap
//      offset.  One b128 read yields the A operand of 4 k-steps (K is
//      permuted so that lane group g owns channels 16h+4g..+3).
//   B (weights): host-packed so that each lane's 8 values per tap are two
//      coalesced 16-byte global loads; streamed L2 -> registers one tap ahead
//      (no LDS, no barrier in the main loop).
// ---------------------------------------------------------------------------
struct ConvArgs {
  const float* in;     // logical origin of item 0
  float* out;
  const float* skip;   // may alias out (in-place residual add)
  const float* wpack;  // [27][2][2][64][4]
  const float* bias;   // [32]
  const uint8_t* valid;  // [nchunks * kChunk]
  long act_stride;
  int XS, plane, R, nchunks;
};

template <bool RELU_IN, bool RELU_OUT, bool ADD_SKIP>
__global__ __launch_bounds__(kConvThreads) void conv32_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int item = blockIdx.x / a.nchunks;
  const int chunk = blockIdx.x - item * a.nchunks;
  const int m0 = chunk * kChunk;
  const float* src = a.in + (size_t)item * a.act_stride;

  // ---- stage chunk + halo into LDS (3 dz segments) ----
  const int R = a.R;
  const int nf4 = R * 8;
#pragma unroll 1
  for (int seg = 0; seg < 3; ++seg) {
    const long p0 = (long)m0 - (a.XS + 1) + (long)(seg - 1) * a.plane;
    const float* s = src + p0 * kFeatures;
    const int row0 = seg * R;
#pragma unroll 4
    for (int e = tid; e < nf4; e += kConvThreads) {
      const int r = e >> 3, q = e & 7;
      float4 v = *reinterpret_cast<const float4*>(s + (size_t)e * 4);
      if (RELU_IN) {
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f);
        v.w = fmaxf(v.w, 0.0f);
      }
      const int row = row0 + r;
      *reinterpret_cast<float4*>(lds + row * 32 + ((q ^ (row & 7)) << 2)) = v;
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nhalf = wave & 1;
  const int tgrp = wave >> 1;
  const int i = lane & 15;
  const int grp = lane >> 4;

  f32x4 acc[kTilesPerWave];
#pragma unroll
  for (int t = 0; t < kTilesPerWave; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rbase = (a.XS + 1) + tgrp * kTilesPerWave * kTile + i;
  const f32x4* wp =
      reinterpret_cast<const f32x4*>(a.wpack) + nhalf * 128 + lane;

  f32x4 b0 = wp[0], b1 = wp[64];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    f32x4 nb0 = b0, nb1 = b1;
    if (tap + 1 < 27) {
      nb0 = wp[(tap + 1) * 256];
      nb1 = wp[(tap + 1) * 256 + 64];
    }
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int tapoff = kz * R + (ky - 1) * a.XS + (kx - 1);
    f32x4 a0[kTilesPerWave], a1[kTilesPerWave];
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t) {
      const int row = rbase + t * kTile + tapoff;
      const int ad = row * 32 + ((grp ^ (row & 7)) << 2);
      a0[t] = *reinterpret_cast<const f32x4*>(lds + ad);
      a1[t] = *reinterpret_cast<const f32x4*>(lds + (ad ^ 16));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][s], b0[s], acc[t],
                                                      0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][s], b1[s], acc[t],
                                                      0, 0, 0);
    }
    b0 = nb0;
    b1 = nb1;
  }

  // ---- epilogue: D[row = grp*4 + r][col = i] -> out[pos][16*nhalf + i] ----
  const int co = nhalf * 16 + i;
  const float bv = a.bias[co];
  float* dst = a.out + (size_t)item * a.act_stride;
  const float* skp = ADD_SKIP ? a.skip + (size_t)item * a.act_stride : nullptr;
#pragma unroll
  for (int t = 0; t < kTilesPerWave; ++t) {
    const int pbase = m0 + (tgrp * kTilesPerWave + t) * kTile + grp * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = pbase + r;
      if (a.valid[p]) {
        float v = acc[t][r] + bv;
        if (RELU_OUT) v = fmaxf(v, 0.0f);
        if (ADD_SKIP) v += skp[(size_t)p * kFeatures + co];
        dst[(size_t)p * kFeatures + co] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// conv32c (conv_variant 2): the exact-f32 conv, "compact + K-split" -- fewer
// MFMAs on the critical path of a single field of view than conv32 above, which
// walks the PADDED position space (6.2 % padding positions are computed and
// dropped) in chunks of 160: 239 of 256 CUs at batch 1, 5 tiles x 27 taps per
// wave = 1,080 MFMAs.  conv32c walks the DENSE FoV index v
// (valid positions only; `pidx[v]` maps it to the padded position) in chunks of
// 144 = 9 tiles -> 250 workgroups, and splits the middle tile's 27 taps between
// the two tile groups: wave (nhalf, tgrp) owns 4 full tiles plus 14 (tgrp 0) or
// 13 (tgrp 1) taps of tile 4 = 122 / 121 tile-taps = 976 MFMAs
// (-9.6 %).  The two partial sums of tile 4 meet in the LDS transpose of the
// epilogue.  The three dz segments of the input are staged progressively (all
// loads in flight from the start; segment kz is written to LDS just before tap
// 9*kz), so only the first third of the staging latency is exposed.
// Lane -> LDS row is no longer affine in the lane id (row ends / plane ends
// insert gaps), so each lane carries the LDS offset of its position per tile.
// ---------------------------------------------------------------------------
constexpr int kCChunk = 144;
constexpr int kCTiles = 9;
// LDS row stride in floats: 32 channels + 8 pad.  With 160-byte rows the
// ds_read_b128 of 16 consecutive rows is bank-conflict free WITHOUT an XOR
// swizzle (brute-forced over the b128 lane groups), so the address of every
// tap is affine: lane base + wave-uniform offset.
constexpr int kCLdsStride = 40;

struct ConvCArgs {
  const float* in;
  float* out;
  const float* skip;
  const float* wpack;
  const float* bias;
  const int32_t* pidx;   // [nchunks_c * 144] dense index -> padded position
  long act_stride;
  int XS, plane, Rc;     // Rc = LDS rows per dz segment (multiple of 32)
  int nchunks, V;
  int fx, fyfx;          // FoV row length and plane size (dense index math)
  int total_slots, slots_per_xcd;
  unsigned nbytes;       // bytes of one activation buffer past its origin
  int store_policy;      // epilogue stores: 0 write-back, 1 sc1, 2 nt
  long long* dbg;        // optional [4 waves][6]: shader / wall clocks of WG 0
  // HEAD instantiation only (fused 1x1x1 head on the last conv of the stack)
  const float* head_w;     // [32] weights + bias
  const float* seed_raw;   // [n][V] raw seed FoV (NaN = never visited)
  float* logits;           // [n][V]
  unsigned* head_count;    // [n * nchunks] per-chunk count of logits >= move_thr
  float pad_value, move_thr;
  // fp16x2 scheme only: *range_flag = range_tag when an operand is outside the
  // fp16 range (the step is then void and re-run with the exact-f32 kernel)
  unsigned* range_flag;
  unsigned range_tag;
};

template <int NT>
__device__ __forceinline__ void mfma_tiles(const f32x4 (&A)[5], const f32x4& B,
                                           f32x4 (&acc)[5]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t][s], B[s], acc[t], 0, 0, 0);
  }
}

// DBG (0 in production): 8 = no A-fragment reads after the first, 16 = no
// weight loads after the first two (issue-rate experiments).
// KS = 16-B staging loads per lane and dz segment: 8 (Rc = 256 rows, e.g. the
// 33^3 FoV) or 9 (Rc = 288).
//
// HEAD (last conv of the stack only): the epilogue does not store the residual
// stream but finishes the network -- ReLU, 1x1x1 conv 32->1 + bias, logits =
// seed + update (convstack_3d.py:51-54,91-94) and this chunk's count of logits
// >= move_threshold -- saving the head launch and 4.6 MB of stores per FoV.
template <bool RELU_IN, bool RELU_OUT, bool ADD_SKIP, int DBG = 0, int KS = 8,
          bool HEAD = false>
__global__ __launch_bounds__(kConvThreads, 2) void conv32c_kernel(ConvCArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const long long dbg_c0 = a.dbg ? clock64() : 0;
  const long long dbg_w0 = a.dbg ? wall_clock64() : 0;
  const int gc = (blockIdx.x & 7) * a.slots_per_xcd + (blockIdx.x >> 3);
  if (gc >= a.total_slots) return;
  const int item = gc / a.nchunks;
  const int chunk = gc - item * a.nchunks;
  const int v0 = chunk * kCChunk;
  const int32_t* pidx = a.pidx + v0;
  // padded position of the chunk's first voxel, by arithmetic (a table lookup
  // here would put one more memory round trip in front of the staging loads)
  int p_first;
  {
    int z = (int)((float)v0 / (float)a.fyfx);
    z -= (z * a.fyfx > v0);
    z += ((z + 1) * a.fyfx <= v0);
    const int rem = v0 - z * a.fyfx;
    int y = (int)((float)rem / (float)a.fx);
    y -= (y * a.fx > rem);
    y += ((y + 1) * a.fx <= rem);
    p_first = __builtin_amdgcn_readfirstlane(z * a.plane + y * a.XS +
                                             (rem - y * a.fx));
  }
  const int p_lo = p_first - (a.XS + 1);  // first staged row of the dz=0 segment
  const float* src = a.in + (size_t)item * a.act_stride;
  const int Rc = a.Rc;

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nhalf = wave & 1;
  const int tgrp = wave >> 1;
  const int i = lane & 15;
  const int grp = lane >> 4;

  // LDS float offset of this lane's position in each of its 5 tiles: tiles 0..3
  // (tgrp 0) / 5..8 (tgrp 1), then the shared tile 4.  (Oldest loads of the
  // kernel: the first A-fragment read needs them.)
  int prow[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int tile = t < 4 ? tgrp * 5 + t : 4;
    prow[t] = (pidx[tile * kTile + i] - p_lo) * kCLdsStride + grp * 4;
  }
  // padded position of this thread's 5 epilogue pieces (also old loads: the
  // residual prefetch below needs them without draining the staging loads)
  // thread -> (position j = (tid >> 3) + 32 k, channel quad tid & 7), k = 0..4
  const int q = tid & 7;
  const int j0 = tid >> 3;
  int pj[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = j0 + 32 * k;
    pj[k] = pidx[(j < kCChunk && v0 + j < a.V) ? j : 0];
  }
  // weight fragments of taps 0 and 1: issued BEFORE the staging loads -- vmcnt
  // retires in order, so a weight load queued behind the staging loads would
  // make the first MFMA wait for all three dz segments.
  struct AFrag { f32x4 h0[5], h1[5]; };
  struct BFrag { f32x4 h0, h1; };
  const f32x4* wp =
      reinterpret_cast<const f32x4*>(a.wpack) + nhalf * 128 + lane;
  auto loadB = [&](int s, BFrag& dst) {
    dst.h0 = wp[s * 256];
    dst.h1 = wp[s * 256 + 64];
  };
  BFrag B0, B1, B2;
  loadB(0, B0);
  loadB(1, B1);

  // ---- staging: all 27 x 16-B loads of the three dz segments in flight at
  // once; segment kz is written to LDS (and waited for) only right before the
  // first tap that reads it, so dz = 0, +1 land behind the MFMAs of dz = -1.
  // Only TWO segment slots exist in LDS (dz = +1 overwrites dz = -1 once every
  // wave is past tap 8): 2 x 256 rows x 160 B = 80 KiB, so two workgroups fit
  // on a CU and fill each other's MFMA issue bubbles / staging / epilogue.
  f32x4 sv[3][KS];  // Rc * 8 == KS * 256 float4 per segment
#pragma unroll
  for (int seg = 0; seg < 3; ++seg) {
    const long p0 = (long)p_lo + (long)(seg - 1) * a.plane;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + p0 * kFeatures);
#pragma unroll
    for (int k = 0; k < KS; ++k) sv[seg][k] = s4[tid + k * kConvThreads];
  }
  auto write_segment = [&](int seg) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int e = tid + k * kConvThreads;
      {
        f32x4 v = sv[seg][k];
        if (RELU_IN) {  // tf.nn.relu in front of conv_a (convstack_3d.py:44)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.0f ? v[c] : 0.0f;
        }
        const int row = (seg & 1) * Rc + (e >> 3);  // slot 0: dz -1, +1; slot 1: dz 0
        *reinterpret_cast<f32x4*>(lds + row * kCLdsStride + (e & 7) * 4) = v;
      }
    }
  };

  write_segment(0);
  __syncthreads();

  // ---- main loop: one step = one tap (two half-taps of 4 k-steps) ----
  //   A fragments (LDS -> VGPR, 10 x ds_read_b128) one tap ahead, ring of 2;
  //   B fragments (L2 -> VGPR, 2 x 16 B)           two taps ahead, ring of 3.
  // Tiles 0..3 of the wave run every tap; the shared tile 4 runs in tile group
  // 0 on the first 5 / 4 / 5 taps of the dz = -1 / 0 / +1 segment (14 taps) and
  // in tile group 1 on the other 13 -- balanced PER SEGMENT, because the
  // segment barriers would otherwise serialise the imbalance (two
  // accumulators, so its 8 MFMAs per tap do not form one dependent chain).
  auto a_off = [&](int s) {  // LDS float offset of tap s (compile-time kz/ky/kx)
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    return ((kz & 1) * Rc + (ky - 1) * a.XS + (kx - 1)) * kCLdsStride;
  };
  auto loadA_tile = [&](int t, int off, AFrag& dst) {
    const float* p = lds + prow[t] + off;
    dst.h0[t] = *reinterpret_cast<const f32x4*>(p);
    dst.h1[t] = *reinterpret_cast<const f32x4*>(p + 16);
  };
  auto loadA = [&](int s, AFrag& dst) {
    const int off = a_off(s);
#pragma unroll
    for (int t = 0; t < 5; ++t) loadA_tile(t, off, dst);
  };
  f32x4 acc[4], acc4a, acc4b;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc4a = acc4b = f32x4{0.f, 0.f, 0.f, 0.f};
  AFrag A0, A1;
  const long long dbg_c1 = a.dbg ? clock64() : 0;
  loadA(0, A0);
  if (DBG & 8) loadA(1, A1);

#define FFN_CGROUP(AH, BH, KS)                                               \
  _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) acc[t_] =                 \
      __builtin_amdgcn_mfma_f32_16x16x4f32(AH[t_][KS], BH[KS], acc[t_], 0,   \
                                           0, 0);                            \
  __builtin_amdgcn_sched_barrier(0);
  // PF: prefetch the next tap's A fragments inside this step (false on the last
  // tap of a dz segment: the next segment is not in LDS yet).
#define FFN_CTAP(S, ACUR, ANEXT, BCUR, BNEXT2, PF)                           \
  {                                                                          \
    const bool pa_ = (PF) && (S) + 1 < 27 && !(DBG & 8);                     \
    const int oa_ = a_off((S) + 1);                                          \
    if (pa_) loadA_tile(0, oa_, ANEXT);                                      \
    FFN_CGROUP(ACUR.h0, BCUR.h0, 0)                                          \
    if (pa_) loadA_tile(1, oa_, ANEXT);                                      \
    FFN_CGROUP(ACUR.h0, BCUR.h0, 1)                                          \
    if (pa_) loadA_tile(2, oa_, ANEXT);                                      \
    FFN_CGROUP(ACUR.h0, BCUR.h0, 2)                                          \
    if (pa_) loadA_tile(3, oa_, ANEXT);                                      \
    FFN_CGROUP(ACUR.h0, BCUR.h0, 3)                                          \
    if (pa_) loadA_tile(4, oa_, ANEXT);                                      \
    FFN_CGROUP(ACUR.h1, BCUR.h1, 0)                                          \
    if ((S) + 2 < 27 && !(DBG & 16)) loadB((S) + 2, BNEXT2);                 \
    FFN_CGROUP(ACUR.h1, BCUR.h1, 1)                                          \
    FFN_CGROUP(ACUR.h1, BCUR.h1, 2)                                          \
    FFN_CGROUP(ACUR.h1, BCUR.h1, 3)                                          \
    if ((tgrp == 0) == (((S) % 9) < (((S) / 9) == 1 ? 4 : 5))) { /* ours */ \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; s_ += 2) {                  \
        acc4a = __builtin_amdgcn_mfma_f32_16x16x4f32(                        \
            ACUR.h0[4][s_], BCUR.h0[s_], acc4a, 0, 0, 0);                    \
        acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(                        \
            ACUR.h0[4][s_ + 1], BCUR.h0[s_ + 1], acc4b, 0, 0, 0);            \
      }                                                                      \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; s_ += 2) {                  \
        acc4a = __builtin_amdgcn_mfma_f32_16x16x4f32(                        \
            ACUR.h1[4][s_], BCUR.h1[s_], acc4a, 0, 0, 0);                    \
        acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(                        \
            ACUR.h1[4][s_ + 1], BCUR.h1[s_ + 1], acc4b, 0, 0, 0);            \
      }                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                     \
    }                                                                        \
  }
  // A ring alternates every tap, B ring has period 3: the pattern repeats
  // every 6 taps.  Taps 8 and 17 end a dz segment.
  FFN_CTAP(0, A0, A1, B0, B2, true)
  FFN_CTAP(1, A1, A0, B1, B0, true)
  FFN_CTAP(2, A0, A1, B2, B1, true)
  FFN_CTAP(3, A1, A0, B0, B2, true)
  FFN_CTAP(4, A0, A1, B1, B0, true)
  FFN_CTAP(5, A1, A0, B2, B1, true)
  FFN_CTAP(6, A0, A1, B0, B2, true)
  FFN_CTAP(7, A1, A0, B1, B0, true)
  FFN_CTAP(8, A0, A1, B2, B1, false)
  write_segment(1);
  __syncthreads();
  if (!(DBG & 8)) loadA(9, A1);
  FFN_CTAP(9, A1, A0, B0, B2, true)
  FFN_CTAP(10, A0, A1, B1, B0, true)
  FFN_CTAP(11, A1, A0, B2, B1, true)
  FFN_CTAP(12, A0, A1, B0, B2, true)
  FFN_CTAP(13, A1, A0, B1, B0, true)
  FFN_CTAP(14, A0, A1, B2, B1, true)
  FFN_CTAP(15, A1, A0, B0, B2, true)
  FFN_CTAP(16, A0, A1, B1, B0, true)
  FFN_CTAP(17, A1, A0, B2, B1, false)
  write_segment(2);
  __syncthreads();
  // ---- per-thread epilogue operands: residual input and bias, fetched once the
  // staging registers of the last segment are free (9 taps of MFMAs cover them)
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + q * 4);
  unsigned ooff[5];
  f32x4 skipv[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = j0 + 32 * k;
    const bool ok = j < kCChunk && v0 + j < a.V;
    const int p = pj[k];
    ooff[k] = ok ? ((unsigned)p * kFeatures + q * 4) * 4u : 0x80000000u;
    skipv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ADD_SKIP)
      skipv[k] = *reinterpret_cast<const f32x4*>(
          a.skip + (size_t)item * a.act_stride + (size_t)p * kFeatures + q * 4);
  }
  if (!(DBG & 8)) loadA(18, A0);
  FFN_CTAP(18, A0, A1, B0, B2, true)
  FFN_CTAP(19, A1, A0, B1, B0, true)
  FFN_CTAP(20, A0, A1, B2, B1, true)
  FFN_CTAP(21, A1, A0, B0, B2, true)
  FFN_CTAP(22, A0, A1, B1, B0, true)
  FFN_CTAP(23, A1, A0, B2, B1, true)
  FFN_CTAP(24, A0, A1, B0, B2, true)
  FFN_CTAP(25, A1, A0, B1, B0, true)
  FFN_CTAP(26, A0, A1, B2, B1, true)
#undef FFN_CTAP
#undef FFN_CGROUP

  const long long dbg_c2 = a.dbg ? clock64() : 0;
  // ---- epilogue: accumulators -> LDS [position j][32 ch]; rows 144..159 hold
  // tgrp 1's partial sums of the shared tile 4 ----
  __syncthreads();
  {
    const int co = nhalf * 16 + i;
    const f32x4 acc4 = acc4a + acc4b;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int tile = t < 4 ? tgrp * 5 + t : 4;
      const int jrow = (t == 4 && tgrp == 1) ? kCChunk + grp * 4
                                             : tile * kTile + grp * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lds[(jrow + r) * 32 + co] = t < 4 ? acc[t][r] : acc4[r];
    }
  }
  __syncthreads();
  unsigned head_above = 0;
  {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    float* obase = a.out + (size_t)item * a.act_stride;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.nbytes, 0x00020000);
    f32x4 hw4 = {0.f, 0.f, 0.f, 0.f};
    float hbias = 0.f;
    if (HEAD) {
      hw4 = *reinterpret_cast<const f32x4*>(a.head_w + q * 4);
      hbias = a.head_w[kFeatures];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int j = j0 + 32 * k;
      const int jr = j < kCChunk ? j : 0;
      f32x4 v = *reinterpret_cast<const f32x4*>(lds + jr * 32 + q * 4);
      if (jr >= 4 * kTile && jr < 5 * kTile)  // shared tile: add the other half
        v += *reinterpret_cast<const f32x4*>(
            lds + (kCChunk + jr - 4 * kTile) * 32 + q * 4);
      v += b4;
      if (RELU_OUT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.0f ? v[c] : 0.0f;
      }
      if (ADD_SKIP) v += skipv[k];
      if (HEAD) {
        // 8 lanes hold the 32 channels of position j: dot with the 1x1x1
        // weights (same association as head_kernel), xor-shuffle reduce
        float partial = fmaxf(v[0], 0.f) * hw4[0];
        partial = __builtin_fmaf(fmaxf(v[1], 0.f), hw4[1], partial);
        partial = __builtin_fmaf(fmaxf(v[2], 0.f), hw4[2], partial);
        partial = __builtin_fmaf(fmaxf(v[3], 0.f), hw4[3], partial);
        partial += __shfl_xor(partial, 1);
        partial += __shfl_xor(partial, 2);
        partial += __shfl_xor(partial, 4);
        bool above = false;
        if (q == 0 && ooff[k] != 0x80000000u) {
          const size_t dv = (size_t)item * a.V + (v0 + j);
          float s = a.seed_raw[dv];
          if (s != s) s = a.pad_value;
          const float lg = s + (partial + hbias);
          a.logits[dv] = lg;
          above = lg >= a.move_thr;
        }
        head_above += (unsigned)__popcll(__ballot(above));  // wave-uniform
        continue;
      }
      // store_policy (A/B switch): 0 write-back, 1 write-through (sc1: no
      // dirty L2 lines left for the kernel boundary to flush), 2 non-temporal
      if (a.store_policy == 1)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v),
                                               rs_out, ooff[k], 0, 16);
      else if (a.store_policy == 2)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v),
                                               rs_out, ooff[k], 0, 2);
      else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v),
                                               rs_out, ooff[k], 0, 0);
    }
  }
  if (HEAD) {  // this chunk's count of logits >= move_thr (summed by faces / paste)
    float* cnt = lds + 160 * 32;  // past the transposed accumulators
    if ((tid & 63) == 0) cnt[tid >> 6] = __uint_as_float(head_above);
    __syncthreads();
    if (tid == 0)
      a.head_count[gc] = __float_as_uint(cnt[0]) + __float_as_uint(cnt[1]) +
                         __float_as_uint(cnt[2]) + __float_as_uint(cnt[3]);
  }
  if (a.dbg && gc == 0 && (tid & 63) == 0) {
    long long* d = a.dbg + (tid >> 6) * 6;
    d[0] = dbg_c0;
    d[1] = dbg_c1;
    d[2] = dbg_c2;
    d[3] = clock64();
    d[4] = dbg_w0;
    d[5] = wall_clock64();
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// ---------------------------------------------------------------------------
// conv32d (conv_variant 6): the split-product conv -- every f32 product carried
// as 3 fp16 products (x ~= hi + 2^-11 res) on v_mfma_f32_32x32x16_f16, f32
// accumulation -- with the 27 taps split over the four waves (each tap's 4 KB
// of weight fragments is fetched by ONE wave and serves all tiles), the operand
// split done ONCE by the producer and the staging done by LDS-DMA.
//
// Its predecessors (conv32w8 / conv32k, removed in ABI 7; history up to commit
// af82310) staged f32 activations through registers and split every value on
// the way into LDS -- 5.3x redundantly (three dz segments of 256-288 rows per
// 144-160 outputs) and with ~9 VALU instructions per value in front of or
// between the MFMAs (3.3 K of conv32k's 14.4 K loop cycles,
// profiles/r02_conv32k_ablations.txt).  Here
//   * every layer WRITES its output already split: "split planes" in HBM,
//       plane cp (0..3: hi of channels 8cp..8cp+7, 4..7: scaled residual of
//       channels 8(cp-4)..) = [padded position] x 16 B, same zero guards / zero
//       padding positions as the f32 layout, same 128 B per position in total;
//     ReLU (conv_a's input, convstack_3d.py:44) is applied by the producer too:
//       conv_a writes  T' = split(relu(conv + b))
//       conv_b writes  X  = conv + b + X (f32 residual stream, planes
//                      [8][position][4 ch]: 16 B per position and plane as
//                      well, so every store instruction of the epilogue writes
//                      one contiguous KiB) and X' = split(relu(X));
//   * a dz segment is then 8 contiguous runs of R x 16 B in HBM and lands in LDS
//     with global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPRs, no
//     VALU): 3 KS DMA instructions per wave, all issued at kernel entry, the
//     dz = -1 segment in front of everything else in the memory queue;
//   * the LDS image is plane-major ([chunk plane][row] x 16 B): the
//     ds_read_b128 of 32 consecutive rows is one contiguous 512 B -- bank
//     conflict free without padding, which is what makes the DMA's lane-linear
//     destination usable;
// Arithmetic, summation order, chunks (160 dense voxels), wave roles (the 27
// taps split 7/7/7/6 over the four waves, + one all-zero tap so that every wave
// runs the same straight-line code): a wave's partial sums of its taps are added
// across the waves in wave order, whatever the tile count.  (A single
// accumulator per tile
// with a 2^11-scaled weight plane was built and measured: one third less
// accumulator read-out, but the cross terms then lose bits against the large
// accumulator, and on the 250^3 fixture the run left the oneDNN / f64
// trajectory at step 430 -- see tests/test_gpu_round2.py -- so it is not used.
// Dependent MFMAs issue back to back at the full rate either way,
// profiles/r02_ubench_mfma_dep.txt.)
// The compiler does not see the DMAs nor the loads of the first four weight
// taps (inline asm), so their s_waitcnt vmcnt are placed by hand.  vmcnt
// retires in order; what a workgroup pulls through its CU's 64 B/clk vector
// memory path per launch (110 KB of activations + 112 KB of weight fragments)
// takes 3.5 K cycles to ISSUE, so only what the first taps need is issued in
// front of the first barrier and the rest rides on the MFMAs of taps 0 and 1:
//     W0 (4) | DMA dz=-1 (KS) | W1 (4)          -> barrier 0: vmcnt(4)
//     tap 0: DMA dz=0 (KS), W2 W3 (8), W4 (4, compiler)
//     tap 1: DMA dz=+1 (KS), W5 (4, compiler)  -> barrier 1: vmcnt(KS + 8)
//     tap 2: W6 (4, compiler); tap 3            -> barrier 2: vmcnt(8)
// (No memory operation of the compiler's precedes a DMA it must not wait for:
// its own vmcnt for such a load would count none of them and drain the queue.)
// ---------------------------------------------------------------------------
constexpr int kDChunk = 160;
constexpr int kDTiles = 5;
constexpr int kDThreads = 256;
constexpr int kDRowB = 144;   // epilogue: row stride of the partial sums in LDS
constexpr int kDTaps = 28;    // 27 + the all-zero tap
constexpr int kDTapBytes = 2 * 2 * 1024;  // weight fragments of one tap (hi, res)

// what changes from one conv of the stack to the next (a launch's own in
// ConvDArgs::L; the resident stack, conv32ps_kernel, derives one per layer)
struct ConvLayer {
  const char* in_sp;     // split planes read (position 0 of plane 0, item 0)
  char* out_sp;          // split planes written (T' or X')
  const char* wpack;     // [28][khalf][plane hi, res][64 lanes][8] fp16 (tap 27 = zeros)
  const float* bias;
  long long* dbg;        // debug_clock: this conv's stamps are recorded
  unsigned flow_wait;    // FLOW: inputs are complete once their tiles' words reach this ...
  unsigned flow_set;     // ... and this conv publishes that
  int flow_wait_on;      // 0: behind a kernel boundary, nothing to wait for
  int layer;             // index of the conv in the stack (flow_trace rows)
};

struct ConvDArgs {
  ConvLayer L;
  float* x_f32;          // residual stream, f32 planes [8][position][4] (position 0 of plane 0)
  long item_bytes;       // bytes per item of an activation buffer (split or f32)
  long sp_plane_bytes;   // positions x 16: one chunk plane of the split layout
  int XS, plane, nchunks, V, fx, fyfx, total_slots, slots_per_xcd;
  unsigned magic_nchunks, magic_fyfx, magic_fx;
  int permuted;          // the FoV is laid out with permuted axes (Geom::oa) ...
  int ds0, ds1, ds2;     // ... one step along z' / y' / x' in the caller's dense order
  unsigned sp_bytes;     // bytes of a split / f32 buffer past position 0 (store range)
  int aoff[4 * 8];       // [wave][j]: LDS byte offset of the wave's j-th tap
  int btap[4 * 8];       // [wave][j]: its tap index (weight fragments)
  const float* head_w;
  const float* seed_raw;
  float* logits;
  unsigned* head_count;
  float pad_value, move_thr;
  unsigned* range_flag;
  unsigned range_tag;
  int dbg_wgs;           // debug_clock 2: every workgroup stamps dbg[24 + 4 blockIdx ..];
                         // 3 (value 2 here): the clock stamps come from tail chunk 0
  // FLOW kernels (section "flagged launches" below): one word per producer
  // workgroup (kFlowStride words apart): the sequence number of the last conv
  // whose outputs for its voxels are complete in memory
  unsigned* flow_flags;
  int flow_n_main;       // main chunks (128 voxels) in front of the tail tiles (32)
  unsigned* flow_err;    // number of polls that gave up (the step is void then)
  int flow_halo;         // dense voxels a 3x3x3 neighbourhood reaches back / ahead
  long long* flow_trace; // debug_clock 4: [workgroup slot][kFlowTraceLayers][8] wall-clock
                         // stamps of every FLOW body (entry, poll done, first barrier,
                         // loop end, stores drained, published), else NULL
  int flow_dbg;          // debug bits: 1 wait for EVERY tile of the FoV; 2 buffer_inv sc1
                         // behind the poll; 4 buffer_wbl2 sc1 in front of the publish
};

constexpr int kDbgMaxWgs = 4096;

// dense index v of this layout -> index in the caller's dense [z][y][x] order
// (logits, seed_raw); the identity unless the axes are permuted
__device__ __forceinline__ int caller_index(const ConvDArgs& a, int v) {
  if (!a.permuted) return v;
  const int z = (int)__umulhi((unsigned)v, a.magic_fyfx);
  const int rem = v - z * a.fyfx;
  const int y = (int)__umulhi((unsigned)rem, a.magic_fx);
  return z * a.ds0 + y * a.ds1 + (rem - y * a.fx) * a.ds2;
}

// debug_clock 2: when and where a workgroup ran -- [start, end] on the 100 MHz
// wall clock, HW_ID (wave / SIMD / CU / SH / SE fields) and XCC_ID
__device__ __forceinline__ void stamp_workgroup(const ConvDArgs& a, const ConvLayer& L, long long t0) {
  if (a.dbg_wgs == 1 && L.dbg && threadIdx.x == 0 &&
      blockIdx.x < (unsigned)kDbgMaxWgs) {
    long long* d = L.dbg + 24 + 4 * (long)blockIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[0] = t0;
    d[1] = wall_clock64();
    d[2] = hw;
    d[3] = xcc;
  }
}

// ---------------------------------------------------------------------------
// Flagged launches (FLOW): the conv chain of a single
// FoV without its kernel boundaries.
//
// A dependent launch costs 1.7 us of boundary + 0.3 of start-up + ~1 us of
// first bytes through an L2 the boundary invalidated -- a third of an 8.75-us
// layer -- because the AQL barrier bit holds the next launch back until the
// LAST workgroup of this one has ended and the caches are flushed.  A FLOW
// launch is queued with the barrier bit cleared (hipExtAnyOrderLaunch): its
// workgroups are dispatched as soon as CU slots are free, stage their weight
// taps (no dependency), and then wait -- not for a boundary but for the 32-voxel
// tiles of the previous layer their own rows come from (+- fy fx + fx + 1 dense
// voxels: 75 words for a 128-voxel chunk of the 33^3 FoV), one word per tile:
// the sequence number of the last conv launch whose outputs for that tile are
// complete.  Placement-independent (G16): every activation store of the
// split-product kernels is an sc1 write-through store already, every storing
// wave drains (vmcnt(0)) before one lane publishes the tile words with sc1
// stores; the consumer polls with relaxed agent-scope loads from ONE wave and
// reads the rows with sc1 loads (LDS-DMA and the residual stream alike).
// Write-after-read is covered by the same words: a tile is overwritten two
// launches later by a workgroup that first waited for every reader of it.
// The arithmetic of a FLOW kernel is its plain kernel's, instruction for
// instruction: same bits.  Every spin is bounded; a poll that gives up voids
// the step through the range flag (the host repeats it without FLOW).
// ---------------------------------------------------------------------------
constexpr unsigned kFlowSpinMax = 1u << 15;
constexpr int kFlowTraceLayers = 64;
// FFN_EXPERIMENTS (tools/build_variant.sh exp -DFFN_EXPERIMENTS=1): the arms the
// rounds' A/B runs selected through engine option flow_debug (bits 1, 2, 4, 32,
// 64, 1024, the sleep selector in bits 8-9) and the debug_clock 4 stamps of
// tools/gpu_flow_trace.py.  The shipped build has none of them: flow_debug keeps
// one bit, 2048 = fault injection (main chunk 3 stops publishing: what a
// producer that is not resident looks like; tests/test_gpu_round5.py).
#ifndef FFN_EXPERIMENTS
#define FFN_EXPERIMENTS 0
#endif
constexpr bool kExp = FFN_EXPERIMENTS != 0;
#ifndef FFN_FLOW_TRACE
#define FFN_FLOW_TRACE FFN_EXPERIMENTS
#endif
constexpr int kFlowFaultBit = 2048;
// FFN_ABLATE (tools/build_variant.sh NAME -DFFN_ABLATE=bits): timing-only builds
// of the resident stack with pieces removed -- the results are WRONG; what each
// piece costs is read off tools/gpu_flow_trace.py.  Bits: 1 publish without the
// drain of the stores; 2 main bodies without the per-tap barriers; 4 without the
// weight ring's DMAs inside the tap loop; 8 without the LDS fragment reads inside
// it; 64 no dz = +1 DMA; 128 main bodies wait for nobody.  (Round 5's table:
// profiles/r05_ablation_resident_stack.txt.)
#ifndef FFN_ABLATE
#define FFN_ABLATE 0
#endif
constexpr int kAbl = FFN_ABLATE;
// the consumer's poll: 1 = one round asks for every producer's word and the
// later rounds only for those still missing; 0 = round 4's form (poll the LAST
// producer's word, then look at all of them once: one more memory round trip
// between the last word's arrival and the first DMA)
#ifndef FFN_POLL_MERGED
#define FFN_POLL_MERGED 1
#endif
constexpr bool kPollMerged = FFN_POLL_MERGED != 0;

// The words: one per PRODUCER (a main chunk of 128 voxels, then the tail tiles of
// 32), 256 bytes apart -- polled words that share a line, or a memory channel,
// with the words other workgroups publish slow both sides down (measured: four
// flag stores per workgroup instead of one, or two polls in flight instead of
// one, cost 10 - 25 % of the step).
constexpr int kFlowStride = 64;  // words between two producers' words

__device__ __forceinline__ int flow_unit(const ConvDArgs& a, int d) {
  const int m = a.flow_n_main * 128;  // (= kMChunk)
  return d < m ? d >> 7 : a.flow_n_main + ((d - m) >> 5);
}

// ONE wave: until every producer of dense voxels [d_lo, d_hi] (clipped to the
// FoV) has published conv L.flow_wait (or a later one).  The producers finish
// roughly in index order (the lower planes lead), so the wave first polls ONE
// word, the last producer's -- one memory transaction per poll -- and then
// looks at all of them once.
__device__ __forceinline__ void flow_wait_tiles(const ConvDArgs& a, const ConvLayer& L,
                                                int d_lo, int d_hi, int lane) {
  typedef FFN_GLOBAL unsigned gu32;
  if (d_lo > a.V - 1 || d_hi < 0) return;
  int lo = flow_unit(a, d_lo < 0 ? 0 : d_lo);
  int hi = flow_unit(a, d_hi > a.V - 1 ? a.V - 1 : d_hi);
  if (kExp && (a.flow_dbg & 1)) {
    lo = 0;
    hi = flow_unit(a, a.V - 1);
  }
  gu32* flags = (gu32*)a.flow_flags;
  gu32* vflag = (gu32*)a.range_flag;
  unsigned spins = 0;
  // A poll that gives up voids the step: the word the faces / paste launch looks
  // at, written so that the other XCDs' polls see it (agent scope) ...
  auto give_up = [&]() {
    if (lane == 0) {
      atomicAdd(a.flow_err, 1u);
      __hip_atomic_store(vflag, a.range_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // ... and a step that is void already is not waited for again: every later
  // poll of the launch that reaches its 256th round looks at that word and
  // leaves (one time-out costs the launch ~25 ms, not one per conv and consumer)
  auto void_already = [&]() {
    return (spins & 255u) == 255u &&
           __hip_atomic_load(vflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
               a.range_tag;
  };
  auto nap = [&]() {
    const int sl = kExp ? (a.flow_dbg >> 8) & 3 : 0;
    if (sl == 0) __builtin_amdgcn_s_sleep(8);
    else if (sl == 1) __builtin_amdgcn_s_sleep(2);
    else if (sl == 2) __builtin_amdgcn_s_sleep(16);
    else __builtin_amdgcn_s_sleep(32);
  };
  if constexpr (kPollMerged) {
    // every lane its own producer's word; a lane whose word has arrived stops
    // asking.  The first round costs one transaction per producer (~20), the
    // later ones only ask for the stragglers (the last producers by index, one
    // to three words) -- and no second look at everything stands between the
    // last word's arrival and the barrier the other waves wait at.
    for (int base = lo; base <= hi; base += 64) {
      const int u = base + lane;
      const int last = base + 63 <= hi ? base + 63 : hi;
      bool pending = u <= hi;
      for (;;) {
        // (no divergent branch: a lane that is done asks for the block's last
        // word along with that word's own lane -- the same transaction)
        const unsigned x = __hip_atomic_load(flags + (long)(pending ? u : last) * kFlowStride,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = pending && (int)(x - L.flow_wait) < 0;
        if (!__any(pending)) break;
        if (++spins > kFlowSpinMax) return give_up();
        if (void_already()) return;
        nap();
      }
    }
    return;
  }
  if (!(kExp && (a.flow_dbg & 64))) {
    for (;;) {  // the last producer's word, every lane the same address
      const unsigned x = __hip_atomic_load(flags + (long)hi * kFlowStride, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(x - L.flow_wait) >= 0) break;
      if (++spins > kFlowSpinMax) return give_up();
      if (void_already()) return;
      nap();
    }
  }
  for (int base = lo; base <= hi; base += 64) {
    const int u = base + lane <= hi ? base + lane : hi;
    for (;;) {
      const unsigned x = __hip_atomic_load(flags + (long)u * kFlowStride, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(x - L.flow_wait) >= 0)) break;
      if (++spins > kFlowSpinMax) return give_up();
      if (void_already()) return;
      nap();
    }
  }
}

// every wave of the workgroup, behind its last activation store: drain, meet,
// then one lane publishes the workgroup's word (first dense voxel v0)
__device__ __forceinline__ long long flow_publish(const ConvDArgs& a, const ConvLayer& L,
                                                  int v0, int tid) {
  typedef FFN_GLOBAL unsigned gu32;
  if constexpr (!(kAbl & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (kExp && (a.flow_dbg & 4))
    asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  const long long t_drained = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int unit = flow_unit(a, v0);
  // (fault injection, flow_debug 2048: producer 3 stays silent after the first conv)
  const bool silent = (a.flow_dbg & kFlowFaultBit) && unit == 3 && L.layer >= 1;
  if (tid == 0 && !silent)
    __hip_atomic_store((gu32*)a.flow_flags + (long)unit * kFlowStride,
                       L.flow_set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t_drained;
}

// debug_clock 4: the stamps of one FLOW body, written when it is over (a store
// among the hand-counted loads would shift their vmcnt)
__device__ __forceinline__ void flow_trace_row(const ConvDArgs& a, const ConvLayer& L,
                                               int gc, const long long (&t)[6]) {
  if (a.flow_trace && threadIdx.x == 0 && L.layer < kFlowTraceLayers) {
    long long* d = a.flow_trace + ((long)gc * kFlowTraceLayers + L.layer) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = t[i];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[6] = hw;
    d[7] = xcc;
  }
}

// one LDS-DMA wave instruction: 64 lanes x 16 B, global (sbase + voff) -> LDS
// (lds_dst + 16 lane); invisible to the compiler's vmcnt bookkeeping
// SC1: an agent-scope load (bypasses this CU's L1, coherent with the sc1
// write-through stores of workgroups on other XCDs): what a FLOW kernel reads
// another RUNNING launch's outputs with.
// NOP: the resident stack spills SGPRs to VGPR lanes, and a base restored by
// v_readlane right in front of this statement is a VALU-written SGPR read by a
// VMEM instruction: 5 wait states the compiler does not insert for inline asm
// (symptom: a wrong chunk in ~0.2 % of the workgroup-layers).  The plain
// kernels' bases come from scalar loads and need none.
template <bool SC1 = false, bool NOP = false>
__device__ __forceinline__ void lds_dma16(const char* sbase, unsigned voff,
                                          unsigned lds_dst) {
#define FFN_DMA16(PRE, POST)                                                    \
  asm volatile(PRE "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" POST \
               :                                                                \
               : "v"(voff), "s"(sbase), "s"(lds_dst)                            \
               : "memory")
  if constexpr (SC1 && NOP) FFN_DMA16("s_nop 2\n\t", " sc1");
  else if constexpr (SC1) FFN_DMA16("", " sc1");
  else if constexpr (NOP) FFN_DMA16("s_nop 2\n\t", "");
  else FFN_DMA16("", "");
#undef FFN_DMA16
}

// a 16-B load the compiler does not count either (waited for by hand)
template <int OFF, bool NOP = false>
__device__ __forceinline__ f16x8 hidden_load16(const char* sbase, unsigned voff) {
  f16x8 d;
  if constexpr (NOP)  // (see lds_dma16: a base fresh from v_readlane)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3"
                 : "=v"(d)
                 : "v"(voff), "s"(sbase), "n"(OFF)
                 : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                 : "=v"(d)
                 : "v"(voff), "s"(sbase), "n"(OFF)
                 : "memory");
  return d;
}

// v ~= hi + 2^-11 res (both fp16), 8 values -> one 16-B hi and one 16-B residual
// fragment; the running maximum of |v| feeds the fp16 range check
__device__ __forceinline__ void split8_fp16(const f32x4& v0, const f32x4& v1,
                                            f16x8& hi, f16x8& res,
                                            unsigned& range_max) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 v = h ? v1 : v0;
    f32x4 vh = v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned mbits = __float_as_uint(v[c]) & 0x7fffffffu;
      range_max = mbits > range_max ? mbits : range_max;
      vh[c] = mbits < 0x38800000u ? 0.0f : v[c];  // |x| < 2^-14: all in the residual
    }
    const f16x4 h4 = __builtin_convertvector(vh, f16x4);
    const f32x4 r1 = (v - __builtin_convertvector(h4, f32x4)) * 2048.0f;
    const f16x4 r4 = __builtin_convertvector(r1, f16x4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      hi[4 * h + c] = h4[c];
      res[4 * h + c] = r4[c];
    }
  }
}

// KIND 0: conv_a (out = split(relu(conv + b)));  KIND 1: conv_b (x = conv + b
// [+ x]; out = split(relu(x)));  HEAD (KIND 1 only): the network's head instead
// of any activation output.
// NT = 32-position tiles per workgroup (chunk = 32 NT dense voxels), R = rows per
// dz segment, WPS = workgroups the kernel is built to co-host per CU (waves per
// SIMD).  (5, 32 KS, 1): one workgroup per CU, the batch-1 form.  (3, 208, 2):
// 96-voxel chunks whose three slots fit in 80 KB, so that TWO workgroups share
// a CU and one's MFMAs run under the other's staging / epilogue -- the same
// arithmetic in the same order, bit-identical results (conv_variant 7).
// (1, 144, 2) with KS = 5: a single 32-voxel tile, the form of conv32mt's tail.
// The workgroup computes the 32 NT dense voxels from v0 of FoV `item`; gc = its
// slot in head_count; aoff_tab = a.aoff or the table of another row count.
template <int KIND, bool ADD_SKIP, int KS, bool HEAD, int NT, int R, int WPS,
          bool FLOW = false>
__device__ __forceinline__ void conv32d_body(const ConvDArgs& a, const ConvLayer& L,
                                             const int item,
                                             const int v0, const int gc,
                                             const int* aoff_tab, const bool dbg_here) {
  typedef f16x8 frag_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  static_assert(NT == 5 || NT == 3 || NT == 1, "tile loop: 5, 3 or 1 tiles");
  static_assert(!FLOW || WPS > 1, "FLOW: the everything-up-front issue order");
  static_assert(4 * KS * 64 >= 8 * R && R % 8 == 0, "KS pieces per wave cover a slot");
  constexpr int kChunkD = 32 * NT;  // dense voxels per workgroup
  constexpr int R16 = R * 16;    // bytes of one chunk plane of a segment in LDS
  constexpr int SEG = 8 * R16;   // bytes of a segment slot
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const long long dbg_c0 = L.dbg ? clock64() : 0;
  const long long dbg_w0 = L.dbg ? wall_clock64() : 0;
  long long ft[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (FLOW) ft[0] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int aoffs[7], btaps[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    aoffs[j] = aoff_tab[wave * 8 + j];
    btaps[j] = a.btap[wave * 8 + j];
  }
  // dense FoV index -> padded position, by arithmetic: a table look-up would be
  // a memory operation of the compiler's in front of the DMAs (see above)
  auto padded = [&](int v) {
    v = v < a.V ? v : a.V - 1;
    const int z = (int)__umulhi((unsigned)v, a.magic_fyfx);
    const int rem = v - z * a.fyfx;
    const int y = (int)__umulhi((unsigned)rem, a.magic_fx);
    return z * a.plane + y * a.XS + (rem - y * a.fx);
  };
  const int p_first = __builtin_amdgcn_readfirstlane(padded(v0));
  const int p_lo = p_first - (a.XS + 1);  // first staged row of the dz = 0 segment

  const int lane = tid & 63;
  const int li = lane & 31;
  const int lh = lane >> 5;

  struct XFragD { frag_t x[2][2]; };  // activations [khalf][plane hi, res]
  struct WFragD { frag_t w[2][2]; };  // weights     [khalf][plane hi, res]
  WFragD W0, W1, W2, W3, W4;
  auto hiddenW = [&](int s, WFragD& dst) {
    const char* b0 = L.wpack + (long)s * kDTapBytes;
    const unsigned vo = (unsigned)lane * 16;
    dst.w[0][0] = hidden_load16<0, FLOW>(b0, vo);
    dst.w[0][1] = hidden_load16<1024, FLOW>(b0, vo);
    dst.w[1][0] = hidden_load16<2048, FLOW>(b0, vo);
    dst.w[1][1] = hidden_load16<3072, FLOW>(b0, vo);
  };
  auto pinW = [&](WFragD& w) {  // "the data is here": consumers stay below
    asm volatile(""
                 : "+v"(w.w[0][0]), "+v"(w.w[0][1]), "+v"(w.w[1][0]),
                   "+v"(w.w[1][1]));
  };
  const frag_t* wp = reinterpret_cast<const frag_t*>(L.wpack) + lane;
  auto loadW = [&](int s, WFragD& dst) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        dst.w[kh][pl] = wp[((s * 2 + kh) * 2 + pl) * 64];
  };

  // ---- staging: 3 x KS LDS-DMA instructions per wave; only dz = -1 and the
  // first two weight taps in front of the first barrier ----
  const unsigned lbase =
      (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsb;
  const char* g0 = L.in_sp + (long)item * a.item_bytes + (long)p_lo * 16;
  unsigned voff[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    // 16-B unit of the segment image (pieces past the slot's end re-copy its
    // first units: the same bytes to the same place)
    int u = 64 * (wave + 4 * k) + lane;
    u = u >= 8 * R ? u - 8 * R : u;
    const int cp = u / R;
    voff[k] = (unsigned)(cp * (int)a.sp_plane_bytes + (u - cp * R) * 16);
  }
  auto dma_piece = [&](int seg, int k) {
    const int u0 = 64 * (wave + 4 * k);  // wave-uniform; wraps with the units
    lds_dma16<FLOW, FLOW>(g0 + (long)(seg - 1) * a.plane * 16, voff[k],
                    lbase + seg * SEG + (u0 >= 8 * R ? u0 - 8 * R : u0) * 16);
  };
  // WPS == 2: a neighbour workgroup's MFMAs cover this one's issue time, so
  // EVERYTHING is queued up front and the later barriers never wait for a DMA
  constexpr bool kEarly = WPS > 1;
  if constexpr (FLOW) {
    // the weights depend on nothing: queued first; the rows of the previous
    // launch only once their tiles are published
    hiddenW(btaps[0], W0);
    hiddenW(btaps[1], W1);
    hiddenW(btaps[2], W2);
    hiddenW(btaps[3], W3);
    if (L.flow_wait_on) {
      if (wave == 0)
        flow_wait_tiles(a, L, v0 - a.flow_halo, v0 + kChunkD - 1 + a.flow_halo, lane);
      ft[1] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kExp && (a.flow_dbg & 2)) asm volatile("buffer_inv sc1" ::: "memory");
    }
#pragma unroll
    for (int seg = 0; seg < 3; ++seg)
#pragma unroll
      for (int k = 0; k < KS; ++k) dma_piece(seg, k);
  } else {
    hiddenW(btaps[0], W0);
#pragma unroll
    for (int k = 0; k < KS; ++k) dma_piece(0, k);
    hiddenW(btaps[1], W1);
    if constexpr (kEarly) {
#pragma unroll
      for (int k = 0; k < KS; ++k) dma_piece(1, k);
      hiddenW(btaps[2], W2);
      hiddenW(btaps[3], W3);
#pragma unroll
      for (int k = 0; k < KS; ++k) dma_piece(2, k);
    }
  }
  // LDS byte offset of this lane's (position, k-group) in each tile
  int xb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    xb[t] = (padded(v0 + t * 32 + li) - p_lo) * 16 + lh * R16;
  // epilogue pieces
  //   normal: item e = tid + 256 k -> (chunk plane c = e / 160, position j = e % 160)
  //   HEAD:   position j = (tid >> 3) + 32 k, channel quad tid & 7
  constexpr int NE = HEAD ? NT : (4 * kChunkD + 255) / 256;
  int ej[NE], ec[NE], ep[NE];
  bool eok[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    if constexpr (HEAD) {
      ej[k] = (tid >> 3) + 32 * k;
      ec[k] = tid & 7;
      eok[k] = v0 + ej[k] < a.V;
    } else {
      const int e = tid + 256 * k;
      ec[k] = e >= 3 * kChunkD ? 3 : e >= 2 * kChunkD ? 2 : e >= kChunkD ? 1 : 0;
      ej[k] = e - kChunkD * ec[k];
      eok[k] = e < 4 * kChunkD && v0 + ej[k] < a.V;
      if (e >= 4 * kChunkD) { ej[k] = 0; ec[k] = 0; }
    }
    ep[k] = padded(v0 + ej[k]);
  }

  auto loadX = [&](int t, int off, XFragD& dst) {
    const char* p = ldsb + xb[t] + off;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
        dst.x[kh][pl] =
            *reinterpret_cast<const frag_t*>(p + (pl * 4 + kh * 2) * R16);
  };
  // acc: products of weight 1 (hi x hi); accC: cross products, weight 2^-11
  f32x16 acc[NT], accC[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = accC[t][r] = 0.f;
  auto mma = [](const frag_t& fw, const frag_t& fx, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, c, 0, 0, 0);
  };
  XFragD X0, X1;

  // W0, dz = -1 landed (newer: W1 [, dz = 0, W2, W3, dz = +1]; FLOW: dz = 0, dz = +1.
  // Its later waits keep the plain order's counts: at least as many operations
  // are newer than what they wait for, the weights landed before the poll)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLOW ? 2 * KS : kEarly ? 2 * KS + 12 : 4)
               : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  pinW(W0);
  const long long dbg_c1 = L.dbg ? clock64() : 0;
  if constexpr (FLOW) ft[2] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  loadX(0, aoffs[0], X0);

  // EXTRA: memory instructions riding on the tile (issued behind its prefetch)
#define FFN_DTILE(T, XCUR, WCUR, PREFETCH, EXTRA)                             \
  __builtin_amdgcn_sched_barrier(0); /* keep the prefetch AHEAD of the MFMAs */ \
  PREFETCH;                                                                   \
  EXTRA;                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                          \
  accC[T] = mma(WCUR.w[0][0], XCUR.x[0][1], accC[T]);                         \
  acc[T] = mma(WCUR.w[0][0], XCUR.x[0][0], acc[T]);                           \
  accC[T] = mma(WCUR.w[0][1], XCUR.x[0][0], accC[T]);                         \
  acc[T] = mma(WCUR.w[1][0], XCUR.x[1][0], acc[T]);                           \
  accC[T] = mma(WCUR.w[1][0], XCUR.x[1][1], accC[T]);                         \
  accC[T] = mma(WCUR.w[1][1], XCUR.x[1][0], accC[T]);
  // tap J of the wave (XA holds tile 0's fragments on entry); CONT: prefetch
  // tile 0 of the next tap under the last tile (false in front of a barrier);
  // E0..E4: the extra memory instructions of its five tiles
#define FFN_DTAP(J, XA, XB, WCUR, CONT, E0, E1, E2, E3, E4)                   \
  {                                                                           \
    const int ao_ = aoffs[J];                                                 \
    const int an_ = aoffs[((J) + 1) % 7];                                     \
    if constexpr (NT == 1) {                                                  \
      (void)ao_;                                                              \
      FFN_DTILE(0, XA, WCUR, if (CONT) loadX(0, an_, XB),                     \
                { E0; E1; E2; E3; E4; })                                      \
    } else {                                                                  \
      FFN_DTILE(0, XA, WCUR, loadX(1, ao_, XB), E0)                           \
      FFN_DTILE(1, XB, WCUR, loadX(2, ao_, XA), E1)                           \
      if constexpr (NT == 3) {                                                \
        FFN_DTILE(2, XA, WCUR, if (CONT) loadX(0, an_, XB), { E2; E3; E4; })  \
      } else {                                                                \
        FFN_DTILE(2, XA, WCUR, loadX(3, ao_, XB), E2)                         \
        FFN_DTILE(3, XB, WCUR, loadX(4, ao_, XA), E3)                         \
        FFN_DTILE(4, XA, WCUR, if (CONT) loadX(0, an_, XB), E4)               \
      }                                                                       \
    }                                                                         \
  }
  auto dma_range = [&](int seg, int k0, int k1) {
    if constexpr (!kEarly) {
#pragma unroll
      for (int k = k0; k < k1 && k < KS; ++k) dma_piece(seg, k);
    }
  };
  auto hiddenW_late = [&](int s, WFragD& dst) {
    if constexpr (!kEarly) hiddenW(s, dst);
  };
  // tap 0: the dz = 0 segment, W2, W3 (hidden), then W4
  FFN_DTAP(0, X0, X1, W0, true, dma_range(1, 0, 3), dma_range(1, 3, 6),
           dma_range(1, 6, KS), hiddenW_late(btaps[2], W2),
           { hiddenW_late(btaps[3], W3); loadW(btaps[4], W4); })
  // W1 landed (newer: dz = 0, W2, W3 [, dz = +1], W4)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEarly ? 2 * KS + 12 : KS + 12)
               : "memory");
  pinW(W1);
  // tap 1: the dz = +1 segment, then W5
  FFN_DTAP(1, X1, X0, W1, false, dma_range(2, 0, 3), dma_range(2, 3, 6),
           dma_range(2, 6, KS), loadW(btaps[5], W0), (void)0)
  // dz = 0, W2, W3 landed: newer are the dz = +1 DMAs and the two compiler taps
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS + 8) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  pinW(W2);
  pinW(W3);
  loadX(0, aoffs[2], X0);
  FFN_DTAP(2, X0, X1, W2, true, loadW(btaps[6], W1), (void)0, (void)0, (void)0,
           (void)0)
  FFN_DTAP(3, X1, X0, W3, false, (void)0, (void)0, (void)0, (void)0, (void)0)
  // dz = +1 landed (newer: W5, W6 [, W4])
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEarly ? 12 : 8) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  loadX(0, aoffs[4], X0);
  // residual input and bias of this thread's epilogue pieces
  f32x4 skipv[NE][2], biasv[NE][2];
  float seedv[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    skipv[k][0] = skipv[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    seedv[k] = 0.f;
    if constexpr (HEAD) {
      if ((tid & 7) == 0 && eok[k])
        seedv[k] = a.seed_raw[(size_t)item * a.V + caller_index(a, v0 + ej[k])];
    }
    if constexpr (HEAD) {
      biasv[k][0] = biasv[k][1] =
          *reinterpret_cast<const f32x4*>(L.bias + (tid & 7) * 4);
    } else {
      biasv[k][0] = *reinterpret_cast<const f32x4*>(L.bias + ec[k] * 8);
      biasv[k][1] = *reinterpret_cast<const f32x4*>(L.bias + ec[k] * 8 + 4);
    }
    if (ADD_SKIP) {
      const float* xs = a.x_f32 + (long)item * (a.item_bytes >> 2);
      if constexpr (FLOW) {
        // the residual stream was written by another launch that may still be
        // running elsewhere: agent-scope loads (the compiler counts these)
        const __amdgpu_buffer_rsrc_t rs_skip = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xs), 0, a.sp_bytes, 0x00020000);
        const unsigned o = (unsigned)((HEAD ? ec[k] : 2 * ec[k]) * (int)a.sp_plane_bytes +
                                      ep[k] * 16);
        skipv[k][0] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_skip, o, 0, 16));
        if constexpr (!HEAD)
          skipv[k][1] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                         rs_skip, o, (int)a.sp_plane_bytes, 16));
      } else if constexpr (HEAD) {
        const int q = ec[k];
        skipv[k][0] = *reinterpret_cast<const f32x4*>(
            xs + (long)q * (a.sp_plane_bytes >> 2) + (long)ep[k] * 4);
      } else {
        const float* s =
            xs + (long)(2 * ec[k]) * (a.sp_plane_bytes >> 2) + (long)ep[k] * 4;
        skipv[k][0] = *reinterpret_cast<const f32x4*>(s);
        skipv[k][1] = *reinterpret_cast<const f32x4*>(s + (a.sp_plane_bytes >> 2));
      }
    }
  }
  FFN_DTAP(4, X0, X1, W4, true, (void)0, (void)0, (void)0, (void)0, (void)0)
  FFN_DTAP(5, X1, X0, W0, true, (void)0, (void)0, (void)0, (void)0, (void)0)
  FFN_DTAP(6, X0, X1, W1, false, (void)0, (void)0, (void)0, (void)0, (void)0)
#undef FFN_DTAP
#undef FFN_DTILE

  const long long dbg_c2 = L.dbg ? clock64() : 0;
  if constexpr (FLOW) ft[3] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  // ---- epilogue: the four waves' partial sums meet in LDS ----
  // P[wave][position 0..159][32 ch] at a 144-B row stride; accumulator register
  // 4 g + i of a lane is channel 8 g + 4 (lane >> 5) + i of position lane & 31.
  __builtin_amdgcn_sched_barrier(0);  // (no accumulator leaves the AGPRs early)
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  {
    char* P = ldsb + wave * (kChunkD * kDRowB) + li * kDRowB + lh * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // tile by tile (the scheduler would otherwise pull every accumulator out
      // of the AGPRs at once and spill the kernel's long-lived values)
      asm volatile("" : "+a"(acc[t]), "+a"(accC[t]));  // still AGPRs here
      const f32x16 s = acc[t] + accC[t] * 4.8828125e-4f;  // 2^-11
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(P + t * (32 * kDRowB) + g * 32) =
            f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  unsigned range_max = 0;
  unsigned head_above = 0;
  if constexpr (HEAD) {
    const int q = tid & 7;
    const f32x4 hw4 = *reinterpret_cast<const f32x4*>(a.head_w + q * 4);
    const float hbias = a.head_w[kFeatures];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int j = ej[k];
      const char* pp = ldsb + j * kDRowB + q * 16;
      f32x4 v = *reinterpret_cast<const f32x4*>(pp);
#pragma unroll
      for (int w = 1; w < 4; ++w)
        v += *reinterpret_cast<const f32x4*>(pp + w * (kChunkD * kDRowB));
      v += biasv[k][0];
      if (ADD_SKIP) v += skipv[k][0];
      float partial = fmaxf(v[0], 0.f) * hw4[0];
      partial = __builtin_fmaf(fmaxf(v[1], 0.f), hw4[1], partial);
      partial = __builtin_fmaf(fmaxf(v[2], 0.f), hw4[2], partial);
      partial = __builtin_fmaf(fmaxf(v[3], 0.f), hw4[3], partial);
      partial += __shfl_xor(partial, 1);
      partial += __shfl_xor(partial, 2);
      partial += __shfl_xor(partial, 4);
      bool above = false;
      if (q == 0 && eok[k]) {
        const size_t dv = (size_t)item * a.V + caller_index(a, v0 + j);
        float s = seedv[k];
        if (s != s) s = a.pad_value;
        const float lg = s + (partial + hbias);
        a.logits[dv] = lg;
        above = lg >= a.move_thr;
      }
      head_above += (unsigned)__popcll(__ballot(above));  // wave-uniform
    }
    float* cnt = reinterpret_cast<float*>(ldsb + 4 * kChunkD * kDRowB);
    if ((tid & 63) == 0) cnt[tid >> 6] = __uint_as_float(head_above);
    __syncthreads();
    if (tid == 0)
      a.head_count[gc] = __float_as_uint(cnt[0]) + __float_as_uint(cnt[1]) +
                         __float_as_uint(cnt[2]) + __float_as_uint(cnt[3]);
  } else {
    const __amdgpu_buffer_rsrc_t rs_sp = __builtin_amdgcn_make_buffer_rsrc(
        L.out_sp + (long)item * a.item_bytes, 0, a.sp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.x_f32) + (long)item * a.item_bytes, 0, a.sp_bytes,
        0x00020000);
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int j = ej[k], c = ec[k];
      const char* pp = ldsb + j * kDRowB + c * 32;
      f32x4 va = *reinterpret_cast<const f32x4*>(pp);
      f32x4 vb = *reinterpret_cast<const f32x4*>(pp + 16);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        va += *reinterpret_cast<const f32x4*>(pp + w * (kChunkD * kDRowB));
        vb += *reinterpret_cast<const f32x4*>(pp + w * (kChunkD * kDRowB) + 16);
      }
      va += biasv[k][0];
      vb += biasv[k][1];
      if (KIND == 1) {
        if (ADD_SKIP) {
          va += skipv[k][0];
          vb += skipv[k][1];
        }
        // the residual stream stays f32 (write-through: nothing dirty is left
        // in L2 for the kernel boundary)
        const unsigned xo = eok[k] ? (unsigned)(2 * c * (int)a.sp_plane_bytes +
                                                ep[k] * 16)
                                   : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, va),
                                               rs_x, xo, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb),
                                               rs_x, xo, (int)a.sp_plane_bytes, 16);
      }
      // what the next conv consumes: ReLU (conv_a's own, or the one in front of
      // the next conv_a), then the split
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        va[cc] = va[cc] > 0.0f ? va[cc] : 0.0f;
        vb[cc] = vb[cc] > 0.0f ? vb[cc] : 0.0f;
      }
      f16x8 hi, res;
      split8_fp16(va, vb, hi, res, range_max);
      const unsigned so = eok[k] ? (unsigned)(c * (int)a.sp_plane_bytes + ep[k] * 16)
                                 : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rs_sp,
                                             so, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, res), rs_sp,
                                             so, (int)(4 * a.sp_plane_bytes), 16);
    }
    // an operand of the next layer left the fp16 range: the step is void, the
    // host re-runs it with the exact-f32 kernel (ffn_step_result.range_error)
    if (!(FLOW && kAbl) && __ballot(range_max > 0x477fe000u) && lane == 0)  // > 65504 (or NaN)
      *a.range_flag = a.range_tag;
    if constexpr (FLOW) ft[4] = flow_publish(a, L, v0, tid);
  }
  if constexpr (FLOW) {
    ft[5] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
    flow_trace_row(a, L, gc, ft);
  }
  if (L.dbg && dbg_here && (tid & 63) == 0) {
    long long* d = L.dbg + wave * 6;
    d[0] = dbg_c0;
    d[1] = dbg_c1;
    d[2] = dbg_c2;
    d[3] = clock64();
    d[4] = dbg_w0;
    d[5] = wall_clock64();
  }
}

template <int KIND, bool ADD_SKIP, int KS, bool HEAD, int NT = 5, int R = 32 * KS,
          int WPS = 1>
__global__ __launch_bounds__(kDThreads, WPS) void conv32d_kernel(ConvDArgs a) {
  const int gc = (blockIdx.x & 7) * a.slots_per_xcd + (blockIdx.x >> 3);
  if (gc >= a.total_slots) return;
  const int item = (int)__umulhi((unsigned)gc, a.magic_nchunks);
  const int chunk = gc - item * a.nchunks;
  conv32d_body<KIND, ADD_SKIP, KS, HEAD, NT, R, WPS>(a, a.L, item, chunk * (32 * NT), gc,
                                                     a.aoff, gc == 0);
}

// ---------------------------------------------------------------------------
// conv32m (conv_variant 8): the same split-product conv, M split over the waves.
//
// conv32d splits K (the taps) over the four waves so that a lone workgroup per CU
// fetches every weight fragment once; the price is the epilogue (four partial
// sums per output meet in LDS: a third of the kernel) and 110 KB of LDS, i.e.
// one workgroup per CU and nothing to run under its prologue and epilogue.
// When several FoVs are in flight there ARE other workgroups, so here
//   * a workgroup = 128 dense voxels, wave w owns tile w (32 positions) for ALL
//     27 taps: no cross-wave reduction, the epilogue goes straight from the
//     accumulators to memory (no LDS, no barrier);
//   * the weights are shared through LDS instead: each tap's 4 KB of fragments
//     is copied ONCE per workgroup by LDS-DMA into a ring of five taps (one
//     1-KiB piece per wave, issued four taps ahead) and read by all four waves;
//   * the waves walk the taps in lock step (one barrier per tap), so the dz = +1
//     segment can take the LDS slot of dz = -1 once every wave is past tap 8:
//     two slots of 240 rows + a ring of five taps = 80 KB, TWO workgroups per
//     CU, <= 156 registers per lane (the accumulators stay in VGPRs: no
//     accumulator read-out) -- one workgroup's MFMAs run under the other's
//     prologue, barriers and epilogue.
// Activations, weights, split planes, staging by DMA, range check, fused head:
// conv32d's.  Every wave accumulates its outputs over all taps in tap order
// (hi x hi, and the two cross products in a second accumulator): the summation
// ORDER differs from conv32d's (partial sums per wave, then added), so the
// logits agree to ~1e-6 but not bit for bit.
// All global loads are inline asm (hidden from the compiler), so every
// s_waitcnt vmcnt is written by hand from the fixed issue order
//   W0 .. W3 | dz=-1 (8) | dz=0 (8) | tap s: W(s+4) [s = 9: dz=+1 (8)]
//   [s = 22: the epilogue operands (NEPI)]
// tap s waits for W(s+1) (prefetched into registers during tap s); the counts
// are computed at compile time from that order (m_wait).
// ---------------------------------------------------------------------------
constexpr int kMChunk = 128;
constexpr int kMRows = 240;
constexpr int kMPieces = 8;                    // DMA pieces per wave and segment
constexpr int kMSeg = 8 * kMRows * 16;         // bytes of a segment slot
constexpr int kMRing = 2 * kMSeg;              // LDS offset of the weight ring
constexpr int kMRingTaps = 5;                  // taps resident in the weight ring
constexpr int kMLdsBytes = kMRing + kMRingTaps * 4096;  // 81,920: two per CU

// vmcnt for tap S's wait (-1: nothing to wait for): operations issued before it
// that are NEWER than W(S+1).  D = ring depth: W0 .. W(D-2) are queued in front
// of the segments, tap t queues W(t+D-1) [t = 9: then the dz = +1 DMAs; t =
// 27 - D: then the NEPI epilogue operands].
// (The dz = +1 segment is queued in ONE tap: spread over taps 9 .. 12 it leaves
// batch 1 unchanged and costs batched steps 5 - 8 %, two workgroups per CU hide
// a one-tap burst better than four taps with a DMA in them:
// profiles/r03_ab_seg_dma_spread_not_kept.txt.)
// FL (FLOW bodies): one more load, the words of the dz = +1 rows' tiles, is
// queued in tap 1 behind its ring piece; it is older than W10, so tap 9's own
// wait covers it.
constexpr int m_wait(int S, int D, int NEPI, bool FL = false) {
  if (S == 0) return kMPieces;      // dz = 0's DMAs are newer than dz = -1 / W1
  if (S + 1 > 26) return -1;
  if (S + 1 <= D - 2) return -1;    // queued in front of everything: landed
  const int tr = S + 2 - D;         // the tap that queued W(S+1)
  int n = 0;
  for (int t = tr; t <= S - 1; ++t) {
    // per tap t, in this order: the ring piece W(t+D-1), the dz = +1 pieces, the
    // epilogue operands
    if (t > tr && t <= 27 - D) n += 1;
    if (FL && t == 1) n += 1;
    if (t == 9) n += kMPieces;
    if (t == 27 - D) n += NEPI;
  }
  return n;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int OFF, bool SC1 = false, bool NOP = false>
__device__ __forceinline__ f32x4 hidden_load16f(const char* sbase, unsigned voff) {
  f32x4 d;
#define FFN_HL16(PRE, POST)                                                  \
  asm volatile(PRE "global_load_dwordx4 %0, %1, %2 offset:%3" POST           \
               : "=v"(d)                                                     \
               : "v"(voff), "s"(sbase), "n"(OFF)                             \
               : "memory")
  if constexpr (SC1 && NOP) FFN_HL16("s_nop 4\n\t", " sc1");
  else if constexpr (SC1) FFN_HL16("", " sc1");
  else if constexpr (NOP) FFN_HL16("s_nop 4\n\t", "");
  else FFN_HL16("", "");
#undef FFN_HL16
  return d;
}

// The workgroup computes the 128 dense voxels from v0 of FoV `item`; gc = its
// slot in head_count.
// RES (the resident stack): the f32 residual stream of the workgroup's voxels
// stays in `xres` (the lane / register layout of the accumulators, the same in
// every conv of the stack) instead of going through memory: conv_b neither
// loads its skip operand nor stores X -- 9.2 MB less per conv_b.
template <int KIND, bool ADD_SKIP, bool HEAD, bool FLOW = false, bool RES = false>
__device__ __forceinline__ void conv32m_body(const ConvDArgs& a, const ConvLayer& L,
                                             const int item,
                                             const int v0, const int gc,
                                             const bool dbg_here,
                                             f32x4* xres = nullptr) {
  typedef f16x8 frag_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int R = kMRows;
  constexpr int R16 = R * 16;
  constexpr bool kSkipLoad = ADD_SKIP && !RES;
  constexpr int NEPI = HEAD ? (kSkipLoad ? 13 : 9) : (kSkipLoad ? 8 : 4);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const long long dbg_c0 = L.dbg ? clock64() : 0;
  const long long dbg_w0 = L.dbg ? wall_clock64() : 0;
  long long ft[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (FLOW) ft[0] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto padded = [&](int v) {
    v = v < a.V ? v : a.V - 1;
    const int z = (int)__umulhi((unsigned)v, a.magic_fyfx);
    const int rem = v - z * a.fyfx;
    const int y = (int)__umulhi((unsigned)rem, a.magic_fx);
    return z * a.plane + y * a.XS + (rem - y * a.fx);
  };
  const int p_first = __builtin_amdgcn_readfirstlane(padded(v0));
  const int p_lo = p_first - (a.XS + 1);
  const int lane = tid & 63;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const unsigned lbase =
      (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsb;

  // ---- weight ring: tap s -> slot s % D, this wave copies piece `wave` ----
  constexpr int D = kMRingTaps;
  auto dma_w = [&](int s) {
    lds_dma16<false, FLOW>(L.wpack + (long)s * kDTapBytes + wave * 1024,
                           (unsigned)lane * 16,
                           lbase + kMRing + (s % D) * 4096 + wave * 1024);
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) dma_w(s);
  // ---- activations: dz = -1 -> slot 0, dz = 0 -> slot 1 (dz = +1 later -> slot 0)
  const char* g0 = L.in_sp + (long)item * a.item_bytes + (long)p_lo * 16;
  unsigned voff[kMPieces];
#pragma unroll
  for (int k = 0; k < kMPieces; ++k) {
    int u = 64 * (wave + 4 * k) + lane;
    u = u >= 8 * R ? u - 8 * R : u;
    const int cp = u / R;
    voff[k] = (unsigned)(cp * (int)a.sp_plane_bytes + (u - cp * R) * 16);
  }
  auto dma_seg = [&](int seg) {  // seg 0, 1, 2 = dz -1, 0, +1
#pragma unroll
    for (int k = 0; k < kMPieces; ++k) {
      const int u0 = 64 * (wave + 4 * k);
      lds_dma16<FLOW, FLOW>(g0 + (long)(seg - 1) * a.plane * 16, voff[k],
                      lbase + (seg & 1) * kMSeg + (u0 >= 8 * R ? u0 - 8 * R : u0) * 16);
    }
  };
  if constexpr (FLOW) {
    // W0 .. W3 are on their way; the rows only once their tiles are published
    if (L.flow_wait_on && !(kAbl & 128)) {
      // the rows of dz = -1 and dz = 0; those of dz = +1 are not needed before
      // tap 9 queues their DMA: their words are fetched during tap 1 (below)
      if (wave == 0)
        flow_wait_tiles(a, L, v0 - a.flow_halo,
                        (kExp && (a.flow_dbg & 32)) ? v0 + kMChunk - 1 + a.flow_halo
                                                    : v0 + kMChunk - 1 + a.fx + 1,
                        lane);
      ft[1] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kExp && (a.flow_dbg & 2)) asm volatile("buffer_inv sc1" ::: "memory");
    }
  }
  dma_seg(0);
  dma_seg(1);

  // this lane's position (its tile = its wave) and its place in the LDS image
  const int jpos = wave * 32 + li;
  const bool ok = v0 + jpos < a.V;
  const int ppos = padded(v0 + jpos);
  const int xb = (ppos - p_lo) * 16 + lh * R16;

  // fragments [khalf][plane hi, res]: the weights of tap s are read one tap
  // ahead (early in tap s - 1: the ring only has them then), the activations
  // TWO taps ahead (three rotating buffers), so that no LDS latency and no
  // straggling read sits between a tap's last MFMA and the next tap's first --
  // with one wave per SIMD nothing else would cover it
  struct XFrag { frag_t x[2][2]; };
  struct WFrag { frag_t w[2][2]; };
  auto load_x = [&](int s, int kh, XFrag& f) {  // 2 of the 4 activation reads of tap s
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    const char* px = ldsb + xb + (kz & 1) * kMSeg + ((ky - 1) * a.XS + (kx - 1)) * 16;
    f.x[kh][0] = *reinterpret_cast<const frag_t*>(px + (0 * 4 + kh * 2) * R16);
    f.x[kh][1] = *reinterpret_cast<const frag_t*>(px + (1 * 4 + kh * 2) * R16);
  };
  auto load_w = [&](int s, int kh, WFrag& f) {  // 2 of the 4 weight reads of tap s
    const char* pw = ldsb + kMRing + (s % D) * 4096 + lane * 16;
    f.w[kh][0] = *reinterpret_cast<const frag_t*>(pw + (kh * 2 + 0) * 1024);
    f.w[kh][1] = *reinterpret_cast<const frag_t*>(pw + (kh * 2 + 1) * 1024);
  };
  f32x16 acc, accC;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = accC[r] = 0.f;
  auto mma = [](const frag_t& fw, const frag_t& fx, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, c, 0, 0, 0);
  };
  // epilogue operands (hidden loads, issued at tap 23)
  f32x4 bias4[4], skip4[4], hw4[4];
  float seedv = 0.f, hbias = 0.f;

  XFrag X0, X1, X2;
  WFrag W0, W1;
  wait_vmcnt<kMPieces>();  // W0 .. W(D-2), dz = -1 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const long long dbg_c1 = L.dbg ? clock64() : 0;
  if constexpr (FLOW) ft[2] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  dma_w(D - 1);
  load_w(0, 0, W0);
  load_w(0, 1, W0);
  load_x(0, 0, X0);
  load_x(0, 1, X0);
  load_x(1, 0, X1);
  load_x(1, 1, X1);

  // tap S: wait for W(S+1), barrier; then the 6 MFMAs of the current fragments
  // with everything else between them, in the shadow of the matrix pipe: the
  // queueing of W(S+D-1) [, the dz = +1 DMAs, the epilogue operands], the 4
  // weight reads of tap S+1 (first: they must be back by its first MFMA) and
  // the 4 activation reads of tap S+2 (nothing waits for them for a whole tap;
  // no lgkmcnt(0) in front of the barrier: every read a ring / segment slot's
  // next DMA could overtake was consumed by an MFMA a tap ago)
  auto dma_seg_part = [&](int k0, int k1) {  // pieces [k0, k1) of dz = +1 -> slot 0
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int u0 = 64 * (wave + 4 * k);
      lds_dma16<FLOW, FLOW>(g0 + (long)a.plane * 16, voff[k],
                      lbase + (u0 >= 8 * R ? u0 - 8 * R : u0) * 16);
    }
  };
  // FLOW: the words of the producers the dz = +1 rows come from.  Wave 0 fetches
  // them during tap 1 (a hidden load, counted in m_wait: the other waves issue a
  // load of the bias line in its place so that every wave's queue has the same
  // length) and looks at them in front of tap 9's barrier, behind which every
  // wave queues its dz = +1 pieces; if a producer has not published yet it polls
  // there while the others wait at the barrier.  (A workgroup's own stores come
  // after every one of its waits, so write-after-read holds as for the eager
  // form.)
  unsigned late_word = 0;
  const int late_d_lo = v0 + a.fyfx - a.fx - 1;
  const bool late_on =
      FLOW && L.flow_wait_on && !(kExp && (a.flow_dbg & 32)) && late_d_lo <= a.V - 1;
  auto flow_late_load = [&]() {
    if constexpr (FLOW) {
      const int lo = flow_unit(
          a, late_d_lo < 0 ? 0 : late_d_lo > a.V - 1 ? a.V - 1 : late_d_lo);
      int hi = v0 + kMChunk - 1 + a.flow_halo;
      hi = flow_unit(a, hi > a.V - 1 ? a.V - 1 : hi);
      const int u = lo + lane <= hi ? lo + lane : hi;
      if (wave == 0)
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1"
                     : "=v"(late_word)
                     : "v"((unsigned)u * (unsigned)(kFlowStride * 4)), "s"(a.flow_flags)
                     : "memory");
      else
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2"
                     : "=v"(late_word)
                     : "v"(0u), "s"(L.bias)
                     : "memory");
    }
  };
  auto flow_late_check = [&]() {
    if constexpr (FLOW) {
      asm volatile("" : "+v"(late_word));
      if (late_on && wave == 0 && !__all((int)(late_word - L.flow_wait) >= 0))
        flow_wait_tiles(a, L, late_d_lo, v0 + kMChunk - 1 + a.flow_halo, lane);
    }
  };
#define FFN_MGAP(S, PART, WNEXT, XNEXT)                                         \
  __builtin_amdgcn_sched_barrier(0);                                            \
  if (!(FLOW && (kAbl & 8)) && (PART) < 2 && (S) + 1 <= 26)                     \
    load_w((S) + 1, PART, WNEXT);                                               \
  if (!(FLOW && (kAbl & 8)) && (PART) >= 2 && (S) + 2 <= 26)                    \
    load_x((S) + 2, (PART) - 2, XNEXT);                                         \
  if (!(FLOW && (kAbl & 64)) && (S) == 9) dma_seg_part(2 * (PART), 2 * (PART) + 2); \
  __builtin_amdgcn_sched_barrier(0);
  // tap S: XCUR / WCUR hold its fragments; WNEXT takes tap S + 1's weights,
  // XNEXT tap S + 2's activations
#define FFN_MTAP(S, XCUR, WCUR, WNEXT, XNEXT)                                   \
  {                                                                             \
    if ((S) > 0) {                                                              \
      if constexpr (m_wait(S, D, NEPI, FLOW) >= 0)                              \
        wait_vmcnt<m_wait(S, D, NEPI, FLOW)>();                                 \
      if (FLOW && !(kAbl & 128) && (S) == 9) flow_late_check();                 \
      if (!(FLOW && (kAbl & 2))) __builtin_amdgcn_s_barrier();                  \
      asm volatile("" ::: "memory");                                            \
    }                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                          \
    accC = mma(WCUR.w[0][0], XCUR.x[0][1], accC);                               \
    __builtin_amdgcn_sched_barrier(0);                                          \
    if (!(FLOW && (kAbl & 4)) && (S) > 0 && (S) + D - 1 <= 26) dma_w((S) + D - 1); \
    if (FLOW && (S) == 1) flow_late_load();                                     \
    __builtin_amdgcn_sched_barrier(0);                                          \
    acc = mma(WCUR.w[0][0], XCUR.x[0][0], acc);                                 \
    FFN_MGAP(S, 0, WNEXT, XNEXT)                                                \
    accC = mma(WCUR.w[0][1], XCUR.x[0][0], accC);                               \
    FFN_MGAP(S, 1, WNEXT, XNEXT)                                                \
    acc = mma(WCUR.w[1][0], XCUR.x[1][0], acc);                                 \
    FFN_MGAP(S, 2, WNEXT, XNEXT)                                                \
    accC = mma(WCUR.w[1][0], XCUR.x[1][1], accC);                               \
    FFN_MGAP(S, 3, WNEXT, XNEXT)                                                \
    accC = mma(WCUR.w[1][1], XCUR.x[1][0], accC);                               \
    __builtin_amdgcn_sched_barrier(0);                                          \
    if ((S) == 27 - D) issue_epilogue_loads();                                  \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }
  auto issue_epilogue_loads = [&]() {
    const unsigned vb = (unsigned)lh * 16;  // channels 8 g + 4 lh .. + 3
    const char* bp = reinterpret_cast<const char*>(L.bias);
    bias4[0] = hidden_load16f<0, false, FLOW>(bp, vb);
    bias4[1] = hidden_load16f<32, false, FLOW>(bp, vb);
    bias4[2] = hidden_load16f<64, false, FLOW>(bp, vb);
    bias4[3] = hidden_load16f<96, false, FLOW>(bp, vb);
    if constexpr (kSkipLoad) {
      // f32 plane 2 g + lh, 16 B per position
      const char* xs = reinterpret_cast<const char*>(a.x_f32) + (long)item * a.item_bytes;
      const unsigned vs = (unsigned)(lh * (int)a.sp_plane_bytes + ppos * 16);
      skip4[0] = hidden_load16f<0, FLOW, FLOW>(xs, vs);
      skip4[1] = hidden_load16f<0, FLOW, FLOW>(xs + 2 * a.sp_plane_bytes, vs);
      skip4[2] = hidden_load16f<0, FLOW, FLOW>(xs + 4 * a.sp_plane_bytes, vs);
      skip4[3] = hidden_load16f<0, FLOW, FLOW>(xs + 6 * a.sp_plane_bytes, vs);
    }
    if constexpr (HEAD) {
      const char* hp = reinterpret_cast<const char*>(a.head_w);
      hw4[0] = hidden_load16f<0, false, FLOW>(hp, vb);
      hw4[1] = hidden_load16f<32, false, FLOW>(hp, vb);
      hw4[2] = hidden_load16f<64, false, FLOW>(hp, vb);
      hw4[3] = hidden_load16f<96, false, FLOW>(hp, vb);
      const char* sp = reinterpret_cast<const char*>(a.seed_raw + (size_t)item * a.V);
      const unsigned so = (unsigned)(caller_index(a, ok ? v0 + jpos : 0) * 4);
      if constexpr (FLOW)  // (lds_dma16: NOP)
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2"
                     : "=v"(seedv)
                     : "v"(so), "s"(sp)
                     : "memory");
      else
        asm volatile("global_load_dword %0, %1, %2"
                     : "=v"(seedv)
                     : "v"(so), "s"(sp)
                     : "memory");
    }
  };
  FFN_MTAP(0, X0, W0, W1, X2)
  FFN_MTAP(1, X1, W1, W0, X0)
  FFN_MTAP(2, X2, W0, W1, X1)
  FFN_MTAP(3, X0, W1, W0, X2)
  FFN_MTAP(4, X1, W0, W1, X0)
  FFN_MTAP(5, X2, W1, W0, X1)
  FFN_MTAP(6, X0, W0, W1, X2)
  FFN_MTAP(7, X1, W1, W0, X0)
  FFN_MTAP(8, X2, W0, W1, X1)
  FFN_MTAP(9, X0, W1, W0, X2)
  FFN_MTAP(10, X1, W0, W1, X0)
  FFN_MTAP(11, X2, W1, W0, X1)
  FFN_MTAP(12, X0, W0, W1, X2)
  FFN_MTAP(13, X1, W1, W0, X0)
  FFN_MTAP(14, X2, W0, W1, X1)
  FFN_MTAP(15, X0, W1, W0, X2)
  FFN_MTAP(16, X1, W0, W1, X0)
  FFN_MTAP(17, X2, W1, W0, X1)
  FFN_MTAP(18, X0, W0, W1, X2)
  FFN_MTAP(19, X1, W1, W0, X0)
  FFN_MTAP(20, X2, W0, W1, X1)
  FFN_MTAP(21, X0, W1, W0, X2)
  FFN_MTAP(22, X1, W0, W1, X0)
  FFN_MTAP(23, X2, W1, W0, X1)
  FFN_MTAP(24, X0, W0, W1, X2)
  FFN_MTAP(25, X1, W1, W0, X0)
  FFN_MTAP(26, X2, W0, W1, X1)
#undef FFN_MTAP
#undef FFN_MGAP
  const long long dbg_c2 = L.dbg ? clock64() : 0;
  if constexpr (FLOW) ft[3] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;

  // ---- epilogue: straight from the accumulators (lane = position jpos,
  // register 4 g + i = channel 8 g + 4 lh + i) ----
  wait_vmcnt<0>();
  asm volatile(""
               : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]));
  if constexpr (kSkipLoad)
    asm volatile(""
                 : "+v"(skip4[0]), "+v"(skip4[1]), "+v"(skip4[2]), "+v"(skip4[3]));
  if constexpr (ADD_SKIP && RES) {
#pragma unroll
    for (int g = 0; g < 4; ++g) skip4[g] = xres[g];
  }
  if constexpr (HEAD)
    asm volatile(""
                 : "+v"(hw4[0]), "+v"(hw4[1]), "+v"(hw4[2]), "+v"(hw4[3]),
                   "+v"(seedv));
  const f32x16 s = acc + accC * 4.8828125e-4f;  // 2^-11
  unsigned range_max = 0;
  if constexpr (HEAD) {
    hbias = a.head_w[kFeatures];
    float partial = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
      v += bias4[g];
      if (ADD_SKIP) v += skip4[g];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        partial = __builtin_fmaf(fmaxf(v[i], 0.f), hw4[g][i], partial);
    }
    partial += __shfl_xor(partial, 32);  // the other 16 channels of the position
    bool above = false;
    if (lh == 0 && ok) {
      const size_t dv = (size_t)item * a.V + caller_index(a, v0 + jpos);
      float sd = seedv;
      if (sd != sd) sd = a.pad_value;
      const float lg = sd + (partial + hbias);
      a.logits[dv] = lg;
      above = lg >= a.move_thr;
    }
    const unsigned mine = (unsigned)__popcll(__ballot(above));
    // (LDS is free: every wave is past its last fragment read only after the
    // barrier below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* cnt = reinterpret_cast<float*>(ldsb);
    if (lane == 0) cnt[wave] = __uint_as_float(mine);
    __syncthreads();
    if (tid == 0)
      a.head_count[gc] = __float_as_uint(cnt[0]) + __float_as_uint(cnt[1]) +
                         __float_as_uint(cnt[2]) + __float_as_uint(cnt[3]);
  } else {
    const __amdgpu_buffer_rsrc_t rs_sp = __builtin_amdgcn_make_buffer_rsrc(
        L.out_sp + (long)item * a.item_bytes, 0, a.sp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.x_f32) + (long)item * a.item_bytes, 0, a.sp_bytes,
        0x00020000);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
      v += bias4[g];
      if (KIND == 1) {
        if (ADD_SKIP) v += skip4[g];
        if constexpr (RES) {
          xres[g] = v;
        } else {
          const unsigned xo =
              ok ? (unsigned)((2 * g + lh) * (int)a.sp_plane_bytes + ppos * 16)
                 : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x,
                                                 xo, 0, 16);
        }
      }
      f32x4 vh;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int bits = __float_as_int(v[cc]);  // ReLU (-0 -> +0, NaN stays)
        v[cc] = __int_as_float(bits > 0 ? bits : 0);
        const unsigned mbits = __float_as_uint(v[cc]);
        range_max = mbits > range_max ? mbits : range_max;
        vh[cc] = mbits < 0x38800000u ? 0.0f : v[cc];  // < 2^-14: all residual
      }
      const f16x4 h4 = __builtin_convertvector(vh, f16x4);
      const f32x4 r1 = (v - __builtin_convertvector(h4, f32x4)) * 2048.0f;
      const f16x4 r4 = __builtin_convertvector(r1, f16x4);
      const unsigned so =
          ok ? (unsigned)(g * (int)a.sp_plane_bytes + ppos * 16 + lh * 8)
             : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rs_sp, so,
                                            0, 16);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r4), rs_sp, so,
                                            (int)(4 * a.sp_plane_bytes), 16);
    }
    if (!(FLOW && kAbl) && __ballot(range_max > 0x477fe000u) && lane == 0)  // > 65504 (or NaN)
      *a.range_flag = a.range_tag;
    if constexpr (FLOW) ft[4] = flow_publish(a, L, v0, tid);
  }
  if constexpr (FLOW) {
    ft[5] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
    flow_trace_row(a, L, gc, ft);
  }
  if (L.dbg && dbg_here && lane == 0) {
    long long* d = L.dbg + wave * 6;
    d[0] = dbg_c0;
    d[1] = dbg_c1;
    d[2] = dbg_c2;
    d[3] = clock64();
    d[4] = dbg_w0;
    d[5] = wall_clock64();
  }
}

template <int KIND, bool ADD_SKIP, bool HEAD>
__global__ __launch_bounds__(kDThreads, 2) void conv32m_kernel(ConvDArgs a) {
  const int gc = (blockIdx.x & 7) * a.slots_per_xcd + (blockIdx.x >> 3);
  if (gc >= a.total_slots) return;
  const long long t0 = a.dbg_wgs ? wall_clock64() : 0;
  const int item = (int)__umulhi((unsigned)gc, a.magic_nchunks);
  const int chunk = gc - item * a.nchunks;
  conv32m_body<KIND, ADD_SKIP, HEAD>(a, a.L, item, chunk * kMChunk, gc, gc == 0);
  stamp_workgroup(a, a.L, t0);
}

// ---------------------------------------------------------------------------
// conv32mt (conv_variant 9): conv32m with a K-split tail.
//
// 256 CUs host two conv32m workgroups each, and a CU that gets two takes the
// matrix-pipe time of both: measured at batch 1, 7.5 us per layer for a FoV of
// <= 256 chunks, 9.75 us for ANY FoV of 257 .. 400 chunks (profiles/
// r02_chunks_vs_cus.txt) -- the 33^3 FoV's 281 chunks pay 30 % for the 25 CUs
// that run two workgroups.  Here the first n_main <= 256 chunks (128 voxels)
// stay conv32m workgroups, one per CU, and the voxels past them go to `tail`
// workgroups of ONE 32-voxel tile whose 27 taps are split over the four waves
// (conv32d's body with a single tile): a tail workgroup that shares a CU adds
// 7 taps, not 27, to each SIMD's matrix work.
// blockIdx -> XCD b & 7 gets mains_per_xcd main chunks FIRST (they take the
// empty CUs), then tails_per_xcd tail chunks of the same region of the FoV.
// The tail sums in conv32d's order (per-wave partial sums, then added), the
// main part in conv32m's: each voxel's arithmetic is fixed by its position in
// the FoV.  conv32d's sums do not depend on its tile count, so a step with
// several FoVs -- where balance over the CUs is no issue but the cost per
// voxel is -- runs the SAME tail voxels in 96-voxel workgroups (TNT = 3,
// conv_variant 7's form) and gets the same bits as a single FoV does.
// ---------------------------------------------------------------------------
constexpr int kTRows = 144;   // TNT = 1: rows per dz segment of a tail workgroup
constexpr int kTPieces = 5;   // its DMA pieces per wave and segment
constexpr int kT3Rows = 208;  // TNT = 3 (= conv_variant 7's kERows / kEPieces)
constexpr int kT3Pieces = 7;

struct ConvTailMap {
  int n;                      // FoVs
  int n_main, n_tail;         // chunks per FoV: 128-voxel main, 32-voxel tail
  int mains_per_xcd, tails_per_xcd;
  int taoff[4 * 8];           // the tail's aoff table (its rows per segment)
};

template <int KIND, bool ADD_SKIP, bool HEAD, int TNT, bool FLOW = false>
__global__ __launch_bounds__(kDThreads, 2) void conv32mt_kernel(ConvDArgs a,
                                                                ConvTailMap mp) {
  static_assert(!FLOW || TNT == 1, "FLOW: the single-FoV form");
  const int xcd = blockIdx.x & 7;
  const int idx = blockIdx.x >> 3;
  int item, r;
  bool main_wg;
  if (TNT == 1) {
    // one FoV at a time: its main chunks first (they take the empty CUs)
    const int per_item = mp.mains_per_xcd + mp.tails_per_xcd;
    item = idx / per_item;
    r = idx - item * per_item;
    main_wg = r < mp.mains_per_xcd;
    if (!main_wg) r -= mp.mains_per_xcd;
  } else {
    // several FoVs: every tail workgroup first -- a K-split workgroup takes
    // longer from start to end than a main one, and started last it would
    // run on alone at the end of the launch
    const int tails = mp.n * mp.tails_per_xcd;
    main_wg = idx >= tails;
    const int i2 = main_wg ? idx - tails : idx;
    const int per = main_wg ? mp.mains_per_xcd : mp.tails_per_xcd;
    item = i2 / per;
    r = i2 - item * per;
  }
  if (item >= mp.n) return;
  const long long t0 = a.dbg_wgs ? wall_clock64() : 0;
  const int slots = mp.n_main + mp.n_tail;
  if (main_wg) {
    const int c = xcd * mp.mains_per_xcd + r;
    if (c >= mp.n_main) return;
    conv32m_body<KIND, ADD_SKIP, HEAD, FLOW>(a, a.L, item, c * kMChunk, item * slots + c,
                                             blockIdx.x == 0 && a.dbg_wgs != 2);
  } else {
    const int c = xcd * mp.tails_per_xcd + r;
    if (c >= mp.n_tail) return;
    constexpr int kPieces = TNT == 1 ? kTPieces : kT3Pieces;
    constexpr int kRows = TNT == 1 ? kTRows : kT3Rows;
    // (everything queued up front, WPS = 2; the staged issue of WPS = 1 -- only
    // W0, dz = -1, W1 in front of the first barrier -- was measured for the
    // single-FoV tail: first barrier at 4.5 K instead of 5.2 K cycles, but the
    // taps 6.9 K instead of 5.8 K: profiles/r02_wg_timeline.txt)
    conv32d_body<KIND, ADD_SKIP, kPieces, HEAD, TNT, kRows, 2, FLOW>(
        a, a.L, item, mp.n_main * kMChunk + c * (32 * TNT), item * slots + mp.n_main + c,
        mp.taoff, item == 0 && c == 0 && a.dbg_wgs == 2);
  }
  stamp_workgroup(a, a.L, t0);
}

// ---------------------------------------------------------------------------
// conv32ps: the whole conv stack of ONE FoV as a single resident launch.
//
// conv32mt's workgroups (256 main chunks, one per CU, + the 32-voxel tail
// workgroups on the CUs' second slots: all resident at once) keep their voxels
// through all 2 depth - 1 convs; between two convs stands, instead of a kernel
// boundary, the FLOW hand-off above: a workgroup starts conv l + 1 as soon as
// the tiles ITS rows come from have been published by conv l.  Each conv's body
// is the plain kernel's (same instructions, same summation order: same bits);
// what changes per conv -- the two activation buffers taking turns, the weights
// and bias of the layer, the sequence numbers -- is derived from the layer
// index.  conv 0 (conv0_b) sits behind the boundary after conv0_a and waits for
// nothing; the last conv carries the fused head and publishes nothing (the
// faces / paste launch behind it is an ordinary dependent launch).
// ---------------------------------------------------------------------------
// (compile-time switch for same-box A/B builds: tools/build_variant.sh)
#ifndef FFN_PS_RES
#define FFN_PS_RES 1
#endif
constexpr bool kPsRes = FFN_PS_RES != 0;

struct ConvStackTab {
  int nlayers;            // 2 depth - 1
  int l_begin, l_end;     // the convs of THIS launch ([0, nlayers) unless debugging)
  int dbg_layer;          // the conv whose clock stamps are recorded (ConvDArgs::L.dbg)
  const char* sp_t;       // T' (position 0 of plane 0): read by even convs, written by odd
  char* sp_s;             // X': written by even convs, read by odd
  const char* wpack0;     // layer 0's weight fragments ...
  long wpack_stride;      // ... bytes per layer
  const float* bias0;
  long bias_stride;       // floats per layer
  unsigned epoch0;        // conv l publishes epoch0 + l + 1 and waits for epoch0 + l
};

__global__ __launch_bounds__(kDThreads, 2) void conv32ps_kernel(ConvDArgs a,
                                                                ConvTailMap mp,
                                                                ConvStackTab tb) {
  const int xcd = blockIdx.x & 7;
  const int r0 = blockIdx.x >> 3;
  const bool main_wg = r0 < mp.mains_per_xcd;
  const int r = main_wg ? r0 : r0 - mp.mains_per_xcd;
  const int c = xcd * (main_wg ? mp.mains_per_xcd : mp.tails_per_xcd) + r;
  if (c >= (main_wg ? mp.n_main : mp.n_tail)) return;
  const long long t0 = a.dbg_wgs ? wall_clock64() : 0;
  const int v0 = main_wg ? c * kMChunk : mp.n_main * kMChunk + c * 32;
  const int gc = main_wg ? c : mp.n_main + c;
  ConvLayer Ldbg = a.L;
  // (experiment, flow_dbg 1024: the main workgroups' waves ahead of the tail's in
  // the CU's arbitration -- a main workgroup that shares its CU with a tail one
  // is what its neighbours wait for)
  if (kExp && (a.flow_dbg & 1024)) {
    if (main_wg) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
  }
  // the residual stream of a main workgroup's voxels (conv32m_body: RES); the
  // tail workgroups keep theirs in memory (their head epilogue has another
  // thread-to-voxel mapping than their conv epilogue)
  f32x4 xres[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xres[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int l = tb.l_begin; l < tb.l_end; ++l) {
    ConvLayer L;
    L.in_sp = (l & 1) ? tb.sp_s : tb.sp_t;
    L.out_sp = (l & 1) ? const_cast<char*>(tb.sp_t) : tb.sp_s;
    L.wpack = tb.wpack0 + (long)l * tb.wpack_stride;
    L.bias = tb.bias0 + (long)l * tb.bias_stride;
    L.dbg = l == tb.dbg_layer ? a.L.dbg : nullptr;
    L.flow_wait = tb.epoch0 + (unsigned)l;
    L.flow_set = tb.epoch0 + (unsigned)l + 1u;
    L.flow_wait_on = l > tb.l_begin;
    L.layer = l;
    if (L.dbg) Ldbg = L;
    const bool last = l == tb.nlayers - 1;
    if (main_wg) {
      const bool dbg_here = blockIdx.x == 0 && a.dbg_wgs != 2;
      if (l == 0)
        conv32m_body<1, false, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else if (last)
        conv32m_body<1, true, true, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else if (l & 1)
        conv32m_body<0, false, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else
        conv32m_body<1, true, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
    } else {
      const bool dbg_here = c == 0 && a.dbg_wgs == 2;
      if (l == 0)
        conv32d_body<1, false, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                    mp.taoff, dbg_here);
      else if (last)
        conv32d_body<1, true, kTPieces, true, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                  mp.taoff, dbg_here);
      else if (l & 1)
        conv32d_body<0, false, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                    mp.taoff, dbg_here);
      else
        conv32d_body<1, true, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                   mp.taoff, dbg_here);
    }
  }
  stamp_workgroup(a, Ldbg, t0);
}

// ---------------------------------------------------------------------------
// head: ReLU -> 1x1x1 conv 32->1 + bias; logits = seed + update
// (reference convstack_3d.py:51-54,91-94; model.py:168-183) and the count of
// logits >= move_threshold that the disco test needs (inference.py:428-431).
// 8 lanes per voxel: one coalesced 128-B line per voxel, xor-shuffle reduce.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(
    const float* __restrict__ X, const float* __restrict__ in_seed,
    float pad_value, const float* __restrict__ wl /*[32] + bias*/,
    float move_thr, float* __restrict__ logits,
    unsigned* __restrict__ block_count /*[n][gridDim.x]*/, Geom g) {
  __shared__ unsigned wave_cnt[4];
  const int item = blockIdx.y;
  const int sub = threadIdx.x & 7;
  const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + sub * 4);
  const float bias = wl[kFeatures];
  unsigned mine = 0;
  for (int v0 = blockIdx.x * 32; v0 < g.V; v0 += gridDim.x * 32) {
    const int v = v0 + (threadIdx.x >> 3);
    float partial = 0.0f;
    const bool live = v < g.V;
    if (live) {
      const int x = v % g.fx;
      const int t = v / g.fx;
      const int y = t % g.fy;
      const int z = t / g.fy;
      const size_t p = (size_t)z * g.plane + y * g.XS + x;
      const f32x4 a = *reinterpret_cast<const f32x4*>(
          X + (size_t)item * g.act_stride + p * kFeatures + sub * 4);
      // max(0, .) is idempotent: correct for raw and pre-activated X
      partial = fmaxf(a[0], 0.f) * w4[0];
      partial = __builtin_fmaf(fmaxf(a[1], 0.f), w4[1], partial);
      partial = __builtin_fmaf(fmaxf(a[2], 0.f), w4[2], partial);
      partial = __builtin_fmaf(fmaxf(a[3], 0.f), w4[3], partial);
    }
    partial += __shfl_xor(partial, 1);
    partial += __shfl_xor(partial, 2);
    partial += __shfl_xor(partial, 4);
    bool above = false;
    if (live && sub == 0) {
      float s = in_seed[(size_t)item * g.V + v];
      if (s != s) s = pad_value;
      const float lg = s + (partial + bias);
      logits[(size_t)item * g.V + v] = lg;
      above = lg >= move_thr;
    }
    mine += (unsigned)__popcll(__ballot(above));  // wave-uniform
  }
  // per-block partial count; the paste kernel sums them (no atomics on one hot
  // address: those serialise at ~12 ns each, and no counter to zero per step)
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0)
    block_count[item * gridDim.x + blockIdx.x] =
        wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// ---------------------------------------------------------------------------
// paste: disco bias + write-back into the canvas seed (inference.py:416-439),
// 6-face max/argmax for the movement policy (movement.py:67-100), and the point
// reads the host queue needs next (inference.py:325,341,503).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float post_disco(float lg, float old, bool disco) {
  // mask = (old < logit(0.5) == 0) & (logits > old); NaN old -> false.
  return (disco && old < 0.0f && lg > old) ? old : lg;
}

__device__ __forceinline__ bool disco_on(unsigned cnt, int V, float thr) {
  // np.mean(bool array) is an f64 division; the threshold is an f32 proto field.
  return thr >= 0.0f && ((double)cnt / (double)V) > (double)thr;
}

__device__ __forceinline__ unsigned sum_block_counts(
    const unsigned* __restrict__ block_count, int head_blocks, int item,
    unsigned* s_cnt /* [blockDim.x / 64] shared */) {
  // total #(logits >= move_thr): sum of the head kernel's per-block partials
  unsigned part = 0;
  for (int e = threadIdx.x; e < head_blocks; e += blockDim.x)
    part += block_count[item * head_blocks + e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = part;
  __syncthreads();
  unsigned cnt = 0;
  for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) cnt += s_cnt[wv];
  return cnt;
}

// #(logits >= move_thr) of the step: the fused head's per-workgroup partials --
// or, when the model's prediction is a centred box of the FoV (Geom::crop), a
// count over that box only (the head counted the whole FoV)
__device__ __forceinline__ unsigned step_count(
    const Geom& g, const float* __restrict__ lg, float move_thr,
    const unsigned* __restrict__ block_count, int head_blocks, int item,
    unsigned* s_cnt) {
  if (!g.crop) return sum_block_counts(block_count, head_blocks, item, s_cnt);
  unsigned part = 0;
  for (int v = threadIdx.x; v < g.V; v += blockDim.x) {
    const int x = v % g.fx, t = v / g.fx;
    part += (in_pred_box(g, t / g.fy, t % g.fy, x) && lg[v] >= move_thr) ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = part;
  __syncthreads();
  unsigned cnt = 0;
  for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) cnt += s_cnt[wv];
  return cnt;
}

// (Measured, round 3: issuing the face / candidate loads and the segmentation ids
// under the faces BEFORE the block-count barrier does not shorten the block --
// 7.1 against 7.0 us for the fused launch; its time is the launch and the two
// PCIe round trips of the publication, not the loads.)
// faces: everything the HOST waits for after a step -- six face max/argmax
// (movement.py:67-100), the point reads of the queue head (inference.py:325,
// 341,503) and the completion flag.  One block per item, launched BEFORE the
// canvas write-back so that the host's queue bookkeeping overlaps the paste.
// Values inside the FoV are recomputed from (logits, old seed) exactly as the
// paste kernel will write them; values outside come from the canvas, which this
// step does not modify there.
constexpr int kPubWords = (int)(sizeof(ffn_step_result) / 4);  // published words

__device__ __forceinline__ void faces_body(
    const int item, const StepItems& si, const Geom& g,
    const float* __restrict__ logits, const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, float deleted_thr, const unsigned* __restrict__ range_flag,
    unsigned range_tag, unsigned long long* __restrict__ pub, unsigned step_id,
    const int* __restrict__ spec_choice, int spec_expected) {
  __shared__ unsigned s_cnt[8];
  __shared__ ffn_step_result s_res;
  const ItemView it = item_view(si, item);
  const float* lg = logits + (size_t)item * g.V;
  const float* old = in_seed + (size_t)item * g.V;
  const unsigned cnt = step_count(g, lg, move_thr, block_count, head_blocks, item, s_cnt);
  const bool disco = disco_on(cnt, g.Vp, disco_thr);
  const int z0 = it.pos[0] - g.fz / 2;
  const int y0 = it.pos[1] - g.fy / 2;
  const int x0 = it.pos[2] - g.fx / 2;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;

  if (wave < 6) {
    const int axis = wave >> 1;
    const int sign = (wave & 1) ? 1 : -1;
    // centre of the prediction (movement.py:60: the centre of `prob_map`)
    const int cz = g.c0[0] + (g.c1[0] - g.c0[0]) / 2;
    const int cy = g.c0[1] + (g.c1[1] - g.c0[1]) / 2;
    const int cx = g.c0[2] + (g.c1[2] - g.c0[2]) / 2;
    // face rows / cols = the two non-fixed axes in zyx order (selects, not
    // runtime-indexed arrays: those would live in scratch memory)
    const int nr = axis == 0 ? 2 * g.dy + 1 : 2 * g.dz + 1;
    const int nc = axis == 2 ? 2 * g.dy + 1 : 2 * g.dx + 1;
    const int total = nr * nc;
    auto dense_index = [&](int e) {
      const int fi = e / nc, fj = e - fi * nc;
      const int z = axis == 0 ? cz + sign * g.dz : cz - g.dz + fi;
      const int y = axis == 1 ? cy + sign * g.dy
                              : (axis == 0 ? cy - g.dy + fi : cy - g.dy + fj);
      const int x = axis == 2 ? cx + sign * g.dx : cx - g.dx + fj;
      return (z * g.fy + y) * g.fx + x;
    };
    float best = -__builtin_inff();
    int besti = 0x7fffffff;
    bool any = false;
    for (int base = 0; base < total; base += 8 * 64) {
      float a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // all loads of the sweep in flight at once
        const int e = base + k * 64 + lane;
        const int v = dense_index(e < total ? e : 0);
        a[k] = lg[v];
        b[k] = old[v];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = base + k * 64 + lane;
        if (e < total) {
          const float val = post_disco(a[k], b[k], disco);
          if (!any || val > best) {  // strict >: first occurrence wins
            best = val;
            besti = e;
            any = true;
          }
        }
      }
    }
    if (!any) {
      best = -__builtin_inff();
      besti = 0x7fffffff;
    }
    // wavefront argmax reduction, ties -> smaller flat index (np.argmax order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(besti, off);
      if (ob > best || (ob == best && oi < besti)) {
        best = ob;
        besti = oi;
      }
    }
    if (lane == 0) {
      s_res.face_score[wave] = best;
      s_res.face_index[wave] = besti;
      int sg = 0;
      if (besti != 0x7fffffff) {
        const int fi = besti / nc, fj = besti - fi * nc;
        const int z = axis == 0 ? cz + sign * g.dz : cz - g.dz + fi;
        const int y = axis == 1 ? cy + sign * g.dy
                                : (axis == 0 ? cy - g.dy + fi : cy - g.dy + fj);
        const int x = axis == 2 ? cx + sign * g.dx : cx - g.dx + fj;
        sg = it.seg[((size_t)(z0 + z) * it.cy + (y0 + y)) * it.cx + (x0 + x)];
      }
      s_res.face_seg[wave] = sg;
    }
  } else if (wave == 6) {
    const int n = it.req->num_candidates;
    if (lane <= n && lane <= FFN_MAX_CANDIDATES) {
      const int32_t* q = lane == 0 ? it.req->start_pos : it.req->candidates[lane - 1];
      const int z = q[0], y = q[1], x = q[2];
      float sv = __builtin_nanf("");
      int gv = 0;
      if (z >= 0 && z < it.cz && y >= 0 && y < it.cy && x >= 0 && x < it.cx) {
        const int lz = z - z0, ly = y - y0, lx = x - x0;
        const size_t ci = ((size_t)z * it.cy + y) * it.cx + x;
        if (in_pred_box(g, lz, ly, lx)) {  // a voxel this step writes
          const int v = (lz * g.fy + ly) * g.fx + lx;
          sv = post_disco(lg[v], old[v], disco);
        } else {
          sv = it.seed[ci];
        }
        gv = it.seg[ci];
      }
      if (lane == 0) {
        s_res.start_logit = sv;
        s_res.num_above_move = cnt;
        s_res.disco_applied = disco ? 1 : 0;
      } else {
        s_res.cand_seed[lane - 1] = sv;
        s_res.cand_seg[lane - 1] = gv;
      }
    }
  } else {
    // keep_history (inference.py:420-423): voxels that were confidently part
    // of the object and that this prediction (before the disco bias) deletes
    unsigned deleted = 0;
    if (deleted_thr == deleted_thr) {  // NaN = not requested
      for (int v = lane; v < g.V; v += 64) {
        const int x = v % g.fx, t = v / g.fx;
        deleted += ((!g.crop || in_pred_box(g, t / g.fy, t % g.fy, x)) &&
                    old[v] >= deleted_thr && lg[v] < 0.0f) ? 1u : 0u;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) deleted += __shfl_xor(deleted, off);
    }
    if (lane == 0) {
      s_res.num_deleted = deleted;
      // (2: this step ran on a speculative conv0_a launch that chose another
      // position than the host did: nothing is pasted, the library repeats it)
      s_res.range_error = (*range_flag == range_tag) ? 1
                          : (spec_expected >= 0 && *spec_choice != spec_expected) ? 2
                                                                                  : 0;
    }
  }
  __syncthreads();
  // Publish from ONE wave, in ONE trip over PCIe: every 32-bit word of the record
  // goes to pinned host memory as an 8-byte word that carries the step number in
  // its upper half (8-byte stores are atomic: a word is either the old step's or
  // this one's), and the host waits until all kPubWords of them carry it.  No
  // record -> system fence -> flag sequence (two more round trips inside the
  // block the next launch waits for).
  if (wave == 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_res);
    unsigned long long* dst = pub + (size_t)item * kPubWords;
    for (int k = lane; k < kPubWords; k += 64)
      __hip_atomic_store(&dst[k], ((unsigned long long)step_id << 32) | src[k],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(512) void faces_kernel(
    StepItems si, Geom g, const float* __restrict__ logits,
    const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, float deleted_thr, const unsigned* __restrict__ range_flag,
    unsigned range_tag, unsigned long long* __restrict__ pub, unsigned step_id,
    const int* __restrict__ spec_choice, int spec_expected) {
  faces_body(blockIdx.x, si, g, logits, in_seed, block_count, head_blocks, move_thr,
             disco_thr, deleted_thr, range_flag, range_tag, pub, step_id,
             spec_choice, spec_expected);
}

// paste: disco bias + write-back into the canvas seed (inference.py:416-439);
// block bx of nbx of FoV `item`.
__device__ __forceinline__ void paste_body(
    const int item, const int bx, const int nbx, const StepItems& si, const Geom& g,
    const float* __restrict__ logits, const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, const unsigned* __restrict__ range_flag, unsigned range_tag,
    const int* __restrict__ spec_choice, int spec_expected) {
  __shared__ unsigned s_cnt[8];
  if (*range_flag == range_tag) return;  // void step (fp16 range): no paste
  // ... or a step whose speculative conv0_a was made for another position
  if (spec_expected >= 0 && *spec_choice != spec_expected) return;
  const ItemView it = item_view(si, item);
  const float* lg = logits + (size_t)item * g.V;
  const float* old = in_seed + (size_t)item * g.V;
  const unsigned cnt = step_count(g, lg, move_thr, block_count, head_blocks, item, s_cnt);
  const bool disco = disco_on(cnt, g.Vp, disco_thr);
  const int z0 = it.pos[0] - g.fz / 2;
  const int y0 = it.pos[1] - g.fy / 2;
  const int x0 = it.pos[2] - g.fx / 2;
  for (int v = bx * blockDim.x + threadIdx.x; v < g.V; v += nbx * blockDim.x) {
    const int x = v % g.fx;
    const int t = v / g.fx;
    const int y = t % g.fy;
    const int z = t / g.fy;
    if (g.crop && !in_pred_box(g, z, y, x)) continue;
    const size_t ci = ((size_t)(z0 + z) * it.cy + (y0 + y)) * it.cx + (x0 + x);
    it.seed[ci] = post_disco(lg[v], old[v], disco);
  }
}

__global__ __launch_bounds__(512) void paste_kernel(
    StepItems si, Geom g, const float* __restrict__ logits,
    const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, const unsigned* __restrict__ range_flag,
    unsigned range_tag, const int* __restrict__ spec_choice, int spec_expected) {
  paste_body(blockIdx.y, blockIdx.x, gridDim.x, si, g, logits, in_seed, block_count,
             head_blocks, move_thr, disco_thr, range_flag, range_tag, spec_choice,
             spec_expected);
}

// A single FoV's faces AND paste as one launch (engine option fuse_paste): block
// 0 is the faces block -- it raises the host's flag as soon as ITS work is done,
// as the separate launch does -- the others paste meanwhile.  Neither reads what
// the other writes (faces recomputes the in-FoV values from the logits), and the
// launch boundary between the two leaves the step's critical path.
__global__ __launch_bounds__(512) void faces_paste_kernel(
    StepItems si, Geom g, const float* __restrict__ logits,
    const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, float deleted_thr, const unsigned* __restrict__ range_flag,
    unsigned range_tag, unsigned long long* __restrict__ pub, unsigned step_id,
    const int* __restrict__ spec_choice, int spec_expected) {
  if (blockIdx.x == 0)
    faces_body(0, si, g, logits, in_seed, block_count, head_blocks, move_thr,
               disco_thr, deleted_thr, range_flag, range_tag, pub, step_id,
               spec_choice, spec_expected);
  else
    paste_body(0, blockIdx.x - 1, gridDim.x - 1, si, g, logits, in_seed, block_count,
               head_blocks, move_thr, disco_thr, range_flag, range_tag, spec_choice,
               spec_expected);
}

// ... and the NEXT step's conv0_a in the same launch (engine option fuse_paste 2,
// the default where a step is followed by a speculative conv0_a): blocks
// kPasteBlocks + 1 .. gather the next FoV from the canvas AS THE PASTE BLOCKS
// NEXT TO THEM ARE LEAVING IT (SeedOverlay: inside this step's prediction box the
// seed is recomputed from the logits, as the faces block does for the queue's
// candidates), so that nothing waits for the paste: one launch and one kernel
// boundary less per step, the conv0_a under the faces' PCIe round trips.  The
// next step's raw seed copy, range flag and choice word are the OTHER of two
// sets (StepSlot): this step's are still being read.
constexpr int kPasteBlocks = 71;
struct Conv0Next {
  float pad_value;
  const float* w;
  const float* bias;
  float* out;
  float* seed_raw;   // the next step's
  Geom q;            // the split-product kernels' layout of the FoV
  int tiles_y, tiles_x;
  Conv0SplitOut so;  // (range flag / tag: the next step's)
  SpecArgs sp;       // (choice: the next step's)
};

__global__ __launch_bounds__(512) void faces_paste_conv0a_kernel(
    StepItems si, Geom g, const float* __restrict__ logits,
    const float* __restrict__ in_seed,
    const unsigned* __restrict__ block_count, int head_blocks, float move_thr,
    float disco_thr, float deleted_thr, const unsigned* __restrict__ range_flag,
    unsigned range_tag, unsigned long long* __restrict__ pub, unsigned step_id,
    const int* __restrict__ spec_choice, int spec_expected, Conv0Next nx) {
  static_assert(kC0Threads == 512, "one block size for the three roles");
  if (blockIdx.x == 0) {
    faces_body(0, si, g, logits, in_seed, block_count, head_blocks, move_thr,
               disco_thr, deleted_thr, range_flag, range_tag, pub, step_id,
               spec_choice, spec_expected);
    return;
  }
  if (blockIdx.x <= kPasteBlocks) {
    paste_body(0, blockIdx.x - 1, kPasteBlocks, si, g, logits, in_seed, block_count,
               head_blocks, move_thr, disco_thr, range_flag, range_tag, spec_choice,
               spec_expected);
    return;
  }
  __shared__ unsigned s_cnt[8];
  const ItemView it = item_view(si, 0);
  SeedOverlay ov;
  // a void step (fp16 range, or a speculative conv0_a made for another position)
  // pastes nothing: the canvas stays as it is
  ov.on = !(*range_flag == range_tag ||
            (spec_expected >= 0 && *spec_choice != spec_expected));
  const unsigned cnt = step_count(g, logits, move_thr, block_count, head_blocks, 0, s_cnt);
  ov.disco = disco_on(cnt, g.Vp, disco_thr) ? 1 : 0;
  ov.lg = logits;
  ov.old = in_seed;
  ov.z0 = it.pos[0] - g.fz / 2;
  ov.y0 = it.pos[1] - g.fy / 2;
  ov.x0 = it.pos[2] - g.fx / 2;
  ov.fy = g.fy;
  ov.fx = g.fx;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    ov.c0[a] = g.c0[a];
    ov.c1[a] = g.c1[a];
  }
  conv0a_body<true>(blockIdx.x - 1 - kPasteBlocks, 0, si, nx.pad_value, nx.w, nx.bias,
                    nx.out, nx.seed_raw, nx.q, nx.tiles_y, nx.tiles_x, nx.so, nx.sp, ov);
}

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

