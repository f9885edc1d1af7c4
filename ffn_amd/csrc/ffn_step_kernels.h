// The kernels around the conv stack of a FoV step: gather + conv0_a in front of it, head / faces / paste / the fused step launches behind it.
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

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
// INLINE_ONLY: a launch that is only ever made for one FoV (the fused step launch) reads the
// kernel-argument copy without looking at use_inline -- one dependent argument load less
template <bool INLINE_ONLY = false>
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
  if (INLINE_ONLY || si.use_inline) {
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

// ov_in.on says whether an overlay MAY apply: its loads are then issued with all the others;
// resolve(ov) is called once, by every thread of the block, when they are in flight, and
// settles ov.on / ov.disco (the fused launch: the step's void flags and its count, whose own
// loads have been in flight since the block started -- one round trip to a memory the launch
// boundary left cold instead of three in a row).
struct Conv0NoResolve {
  __device__ void operator()(SeedOverlay&) const {}
};
template <bool SPLIT, class Resolve = Conv0NoResolve>
__device__ __forceinline__ void conv0a_body(
    const int tile_block, const int item, const StepItems& si, float pad_value,
    const float* __restrict__ w /*[27][2][32]*/,
    const float* __restrict__ bias, float* __restrict__ out,
    float* __restrict__ seed_raw, const Geom& g, int tiles_y, int tiles_x,
    const Conv0SplitOut& so, const SpecArgs& sp, const SeedOverlay& ov_in,
    long long* tr = nullptr, Resolve resolve = Resolve()) {
  SeedOverlay ov = ov_in;
  // (tr: debug_fused_trace stamps, last of every eighth block to get there: [16] position
  // chosen, [17] tile staged in LDS, [18] MFMAs done -- a sample, so that the stamps'
  // atomics do not stand in the way of what they time)
  auto stamp = [&](int k) {
    if (tr && threadIdx.x == 0 && (tile_block & 7) == 0)
      atomicMax(reinterpret_cast<unsigned long long*>(tr + k), (unsigned long long)wall_clock64());
  };
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

  // B fragments: k = 4 s + grp -> w[k][16 nhalf + i]; k >= 54 is zero padding
  // (loaded in front of everything that is waited for: they depend on nothing)
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
    if (tr && threadIdx.x == 0 && tile_block == 0) tr[20] = wall_clock64();  // (loads issued)
    resolve(ov);
    if (tr && threadIdx.x == 0 && tile_block == 0) tr[21] = wall_clock64();  // (count known)
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k)
      if (ov.on && ovi[k] >= 0) sv[k] = post_disco(ol[k], oo[k], ov.disco != 0);
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k)  // (no short-circuit into dependent loads)
      asm volatile("" : "+v"(sv[k]), "+v"(gv[k]));
    int ch = -1;
#pragma unroll
    for (int k = kSpecMax - 1; k >= 0; --k)
      if (k < sp.n && !(sv[k] < sp.move_thr) && gv[k] <= 0) ch = k;
    if (tile_block == 0 && threadIdx.x == 0) *sp.choice = ch;
    if (ch < 0) return;
    stamp(16);
    if (tr && threadIdx.x == 0 && tile_block == 0) tr[22] = wall_clock64();
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
    resolve(ov);
  }

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
  stamp(17);

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
  stamp(18);
  if constexpr (SPLIT) {
    __syncthreads();
    unsigned range_max = 0;
    char* ob = so.out_sp + (long)item * so.item_bytes;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0x7fffffff, 0x00020000);
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
      // write-through (sc1), as the split-product kernels' own epilogues: no dirty L2
      // lines for the launch boundary in front of the stack to write back (4.6 MB here:
      // +0.5 us of boundary, tools/probes/boundary_probe.hip)
      typedef unsigned u32x4_c0 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(u32x4_c0, hi), rs_out,
          (unsigned)((long)c * so.sp_plane_bytes + p * 16), 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(u32x4_c0, res), rs_out,
          (unsigned)((long)(4 + c) * so.sp_plane_bytes + p * 16), 0, 16);
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
    const int* __restrict__ spec_choice, int spec_expected, long long* tr = nullptr) {
  // (tr: debug_fused_trace stamps [12] count known, [13] faces reduced, [14] record built)
  __shared__ unsigned s_cnt[8];
  __shared__ ffn_step_result s_res;
  const ItemView it = item_view(si, item);
  const float* lg = logits + (size_t)item * g.V;
  const float* old = in_seed + (size_t)item * g.V;
  // Every global load of the block is in flight before any is waited for: the block
  // counts, the face values (logits, old seed, segmentation ids under the faces) and the
  // candidates' point values are ONE round trip to a memory the launch boundary left cold
  // (3.5 + 2.6 + 0.9 us when they follow each other: profiles/r06_step_roles.txt), and the
  // host's turn-around starts when this block's record leaves.
  const int z0 = it.pos[0] - g.fz / 2;
  const int y0 = it.pos[1] - g.fy / 2;
  const int x0 = it.pos[2] - g.fx / 2;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  unsigned cnt_part = 0;
  if (!g.crop)
    for (int e = threadIdx.x; e < head_blocks; e += blockDim.x)
      cnt_part += block_count[item * head_blocks + e];
  // face geometry of this wave (waves 0 .. 5)
  const int f_axis = wave >> 1;
  const int f_sign = (wave & 1) ? 1 : -1;
  const int f_cz = g.c0[0] + (g.c1[0] - g.c0[0]) / 2;
  const int f_cy = g.c0[1] + (g.c1[1] - g.c0[1]) / 2;
  const int f_cx = g.c0[2] + (g.c1[2] - g.c0[2]) / 2;
  const int f_nr = f_axis == 0 ? 2 * g.dy + 1 : 2 * g.dz + 1;
  const int f_nc = f_axis == 2 ? 2 * g.dy + 1 : 2 * g.dx + 1;
  const int f_total = f_nr * f_nc;
  auto face_zyx = [&](int e, int& z, int& y, int& x) {
    const int fi = e / f_nc, fj = e - fi * f_nc;
    z = f_axis == 0 ? f_cz + f_sign * g.dz : f_cz - g.dz + fi;
    y = f_axis == 1 ? f_cy + f_sign * g.dy
                    : (f_axis == 0 ? f_cy - g.dy + fi : f_cy - g.dy + fj);
    x = f_axis == 2 ? f_cx + f_sign * g.dx : f_cx - g.dx + fj;
  };
  constexpr int kFaceSweep = 8;  // elements per lane and sweep
  const bool one_sweep = f_total <= kFaceSweep * 64;
  float fa[kFaceSweep], fb[kFaceSweep];
  int fs[kFaceSweep];
  if (wave < 6 && one_sweep) {
#pragma unroll
    for (int k = 0; k < kFaceSweep; ++k) {
      const int e = k * 64 + lane;
      int z, y, x;
      face_zyx(e < f_total ? e : 0, z, y, x);
      const int v = (z * g.fy + y) * g.fx + x;
      fa[k] = lg[v];
      fb[k] = old[v];
      fs[k] = it.seg[((size_t)(z0 + z) * it.cy + (y0 + y)) * it.cx + (x0 + x)];
    }
  }
  // the candidates' point values (wave 6)
  float c_lg = 0.f, c_old = 0.f, c_seed = 0.f;
  int c_seg = 0, c_kind = 0;  // 1: a voxel this step writes, 2: outside its box
  if (wave == 6) {
    const int n = it.req->num_candidates;
    if (lane <= n && lane <= FFN_MAX_CANDIDATES) {
      const int32_t* q = lane == 0 ? it.req->start_pos : it.req->candidates[lane - 1];
      const int z = q[0], y = q[1], x = q[2];
      if (z >= 0 && z < it.cz && y >= 0 && y < it.cy && x >= 0 && x < it.cx) {
        const int lz = z - z0, ly = y - y0, lx = x - x0;
        const size_t ci = ((size_t)z * it.cy + y) * it.cx + x;
        if (in_pred_box(g, lz, ly, lx)) {
          const int v = (lz * g.fy + ly) * g.fx + lx;
          c_lg = lg[v];
          c_old = old[v];
          c_kind = 1;
        } else {
          c_seed = it.seed[ci];
          c_kind = 2;
        }
        c_seg = it.seg[ci];
      }
    }
  }
  if (tr && threadIdx.x == 0) tr[19] = wall_clock64();  // (every load of the block issued)
  unsigned cnt;
  if (!g.crop) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt_part += __shfl_xor(cnt_part, off);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt_part;
    __syncthreads();
    cnt = 0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) cnt += s_cnt[wv];
  } else {
    cnt = step_count(g, lg, move_thr, block_count, head_blocks, item, s_cnt);
  }
  if (tr && threadIdx.x == 0) tr[12] = wall_clock64();
  const bool disco = disco_on(cnt, g.Vp, disco_thr);

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
        if (one_sweep) {  // (loaded in front of the block-count barrier)
          a[k] = fa[k];
          b[k] = fb[k];
          continue;
        }
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
    if (tr && threadIdx.x == 0) tr[13] = wall_clock64();
    if (lane == 0) {
      s_res.face_score[wave] = best;
      s_res.face_index[wave] = besti;
      int sg = 0;
      if (besti != 0x7fffffff && !one_sweep) {
        const int fi = besti / nc, fj = besti - fi * nc;
        const int z = axis == 0 ? cz + sign * g.dz : cz - g.dz + fi;
        const int y = axis == 1 ? cy + sign * g.dy
                                : (axis == 0 ? cy - g.dy + fi : cy - g.dy + fj);
        const int x = axis == 2 ? cx + sign * g.dx : cx - g.dx + fj;
        sg = it.seg[((size_t)(z0 + z) * it.cy + (y0 + y)) * it.cx + (x0 + x)];
      }
      s_res.face_seg[wave] = sg;
    }
    if (one_sweep) {
      // the id under the arg-max: element besti sits in register besti >> 6 of lane
      // besti & 63
      int mine = 0;
#pragma unroll
      for (int k = 0; k < kFaceSweep; ++k) mine = (besti >> 6) == k ? fs[k] : mine;
      const int sgv = __shfl(mine, besti == 0x7fffffff ? 0 : (besti & 63));
      if (lane == 0) s_res.face_seg[wave] = besti == 0x7fffffff ? 0 : sgv;
    }
  } else if (wave == 6) {
    const int n = it.req->num_candidates;
    if (lane <= n && lane <= FFN_MAX_CANDIDATES) {
      // (a voxel this step writes: recomputed as the paste will write it)
      const float sv = c_kind == 1   ? post_disco(c_lg, c_old, disco)
                       : c_kind == 2 ? c_seed
                                     : __builtin_nanf("");
      const int gv = c_seg;
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
      // (3: the void came from the resident launch giving up on a producer --
      // range_flag[1] carries this step's tag then -- not from the range check)
      s_res.range_error = (*range_flag == range_tag) ? (range_flag[1] == range_tag ? 3 : 1)
                          : (spec_expected >= 0 && *spec_choice != spec_expected) ? 2
                                                                                  : 0;
    }
  }
  __syncthreads();
  if (tr && threadIdx.x == 0) tr[14] = wall_clock64();
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
// (how many: the host picks them so that the launch has one block per CU -- 1 faces block +
// paste blocks + conv0_a tiles = the CU count where that leaves at least kPasteBlocksMin -- a
// conv0_a block that shares its CU with another block reaches its position choice 2.7 us
// after the others, and the next stack waits for the last of them)
constexpr int kPasteBlocks = 71;
constexpr int kPasteBlocksMin = 16;
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
    const int* __restrict__ spec_choice, int spec_expected, Conv0Next nx,
    long long* __restrict__ stamps, int trace, int paste_blocks) {
  static_assert(kC0Threads == 512, "one block size for the three roles");
  const long long t_in = wall_clock64();
  warm_kernargs<832>();
  const long long t_warm = wall_clock64();
  // engine option debug_fused_trace (trace != 0): when each role of this launch ran --
  // first entry (min over the blocks) [4] faces, [5] paste, [6] conv0_a; last end (max) [8]
  // faces = record published, [9] paste, [10] conv0_a; ([7], [11]: the stack before it)
  const bool tr_on = stamps && trace != 0;
  struct RoleStamp {
    long long* lo;
    long long* hi;
    bool on;
    // (every eighth block of a role stamps: the stamps' atomics on one word serialise)
    __device__ RoleStamp(long long* l, long long* h, bool o)
        : lo(l), hi(h), on(o && (blockIdx.x <= 1 || (blockIdx.x & 7) == 0)) {
      if (on && threadIdx.x == 0)
        atomicMin(reinterpret_cast<unsigned long long*>(lo), (unsigned long long)wall_clock64());
    }
    __device__ ~RoleStamp() {
      if (on && threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(hi), (unsigned long long)wall_clock64());
    }
  };
  if (blockIdx.x == 0) {
    RoleStamp rs(stamps + 4, stamps + 8, tr_on);
    faces_body(0, si, g, logits, in_seed, block_count, head_blocks, move_thr,
               disco_thr, deleted_thr, range_flag, range_tag, pub, step_id,
               spec_choice, spec_expected, tr_on ? stamps : nullptr);
    // (stat_turn_*: when this step's record left for the host -- the next resident
    // launch measures how long the GPU then waited for it, conv32ps_kernel)
    if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();
    return;
  }
  if ((int)blockIdx.x <= paste_blocks) {
    RoleStamp rs(stamps + 5, stamps + 9, tr_on);
    paste_body(0, blockIdx.x - 1, paste_blocks, si, g, logits, in_seed, block_count,
               head_blocks, move_thr, disco_thr, range_flag, range_tag, spec_choice,
               spec_expected);
    return;
  }
  RoleStamp rs(stamps + 6, stamps + 10, tr_on);
  if (tr_on && threadIdx.x == 0) {  // ([23] last sampled block's entry, [24] block 0: arguments in)
    if (((blockIdx.x - 1 - paste_blocks) & 7) == 0)
      atomicMax(reinterpret_cast<unsigned long long*>(stamps + 23), (unsigned long long)t_in);
    if ((int)blockIdx.x == 1 + paste_blocks) stamps[24] = t_warm;
  }
  __shared__ unsigned s_cnt[8];
  const ItemView it = item_view<true>(si, 0);
  // What decides how the canvas looks once this step has pasted -- its void flags (fp16
  // range / a speculative conv0_a made for another position: it pastes nothing, the canvas
  // stays as it is) and its count (the disco test) -- is LOADED here, without a branch and
  // without a look at the values, and looked at when the conv0_a body has its own loads in
  // flight (resolve): per-lane loads, so that no scalar wait of the body's stands behind
  // them.  (A loop or a compare here and the compiler waits for the load on the spot: each
  // of those was a cold round trip of its own, profiles/r06_step_roles.txt.)
  int z = 0;
  asm volatile("" : "+v"(z));
  unsigned rf = range_flag[z];
  int sc = spec_choice[z];
  const int e0 = (int)threadIdx.x < head_blocks ? (int)threadIdx.x : 0;
  unsigned part = block_count[e0];
  SeedOverlay ov;
  ov.on = 1;
  ov.disco = 0;
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
  auto resolve = [&](SeedOverlay& o) {
    // (the values are looked at from here on: everything above has been issued)
    asm volatile("" : "+v"(rf), "+v"(sc), "+v"(part) : : "memory");
    unsigned cnt;
    if (g.crop) {
      cnt = step_count(g, logits, move_thr, block_count, head_blocks, 0, s_cnt);
    } else {
      unsigned p2 = (int)threadIdx.x < head_blocks ? part : 0u;
      for (int e = kC0Threads + (int)threadIdx.x; e < head_blocks; e += kC0Threads)
        p2 += block_count[e];  // (more partials than threads: no geometry has them today)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p2 += __shfl_xor(p2, off);
      if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = p2;
      __syncthreads();
      cnt = 0;
#pragma unroll
      for (int wv = 0; wv < kC0Threads / 64; ++wv) cnt += s_cnt[wv];
    }
    o.on = !(rf == range_tag || (spec_expected >= 0 && sc != spec_expected));
    o.disco = disco_on(cnt, g.Vp, disco_thr) ? 1 : 0;
  };
  conv0a_body<true>(blockIdx.x - 1 - paste_blocks, 0, si, nx.pad_value, nx.w, nx.bias,
                    nx.out, nx.seed_raw, nx.q, nx.tiles_y, nx.tiles_x, nx.so, nx.sp, ov,
                    tr_on ? stamps : nullptr, resolve);
}

}  // namespace ffn
