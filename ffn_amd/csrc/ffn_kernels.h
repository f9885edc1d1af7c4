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

// The kernel arguments of a launch that has just crossed a boundary are cold (the scalar
// cache was invalidated, the host wrote them to device memory), and a kernel with roles and
// branches reads them in STAGES: every s_load behind a branch is its own miss, 0.5 - 1 us
// each, one behind the other (the fused step launch: six stages before its first data load).
// This touches every 64-byte line of the first BYTES of the segment at once and waits for
// them once: one miss, the stages after it hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned sink;
  static_assert(BYTES > 0 && BYTES <= 1024, "");
  // (line L is touched when the segment reaches it: assembler conditionals on BYTES)
#define FFN_KA(OFF) ".if %2 > " #OFF "\n\ts_load_dword %0, %1, " #OFF "\n\t.endif\n\t"
  asm volatile(FFN_KA(0x0) FFN_KA(0x40) FFN_KA(0x80) FFN_KA(0xc0) FFN_KA(0x100) FFN_KA(0x140)
               FFN_KA(0x180) FFN_KA(0x1c0) FFN_KA(0x200) FFN_KA(0x240) FFN_KA(0x280)
               FFN_KA(0x2c0) FFN_KA(0x300) FFN_KA(0x340) FFN_KA(0x380) FFN_KA(0x3c0)
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(sink) : "s"(ka), "n"(BYTES) : "memory");
#undef FFN_KA
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

}  // namespace ffn

// The kernels, in dependency order (each header opens namespace ffn itself):
#include "ffn_step_kernels.h"    // conv0_a in front of the stack; head, faces, paste behind it
#include "ffn_conv_exact.h"      // conv_variant 0, 2: exact f32
#include "ffn_conv_split.h"      // conv_variant 6 .. 9: fp16 split products, FLOW hand-off
#include "ffn_conv_resident.h"   // conv32ps: the stack as one launch
#include "ffn_conv_half.h"       // conv32h / conv32hs: 64-voxel workgroups, two chains per SIMD
#include "ffn_canvas_kernels.h"  // box I/O, commit, segment turn
