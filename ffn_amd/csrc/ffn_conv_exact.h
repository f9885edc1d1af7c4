// The exact-f32 3x3x3 32->32 convs (conv_variant 0 and 2): v_mfma_f32_16x16x4_f32, bitwise the oracle's fmaf chain.
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

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

}  // namespace ffn
