// conv32h / conv32hs: the split-product conv on 16-position tiles, 80-voxel workgroups -- two
// independent hand-off chains per SIMD inside ONE FoV (conv_variant 10; measured, not the default).
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

// ---------------------------------------------------------------------------
// Why.  conv32ps (ffn_conv_resident.h) runs one wave per SIMD: a layer's 2.56 us
// of MFMAs sit inside a 7.2-us chain  words seen -> rows staged -> taps ->
// stores drained -> word published  with nothing else to run on the SIMD while
// the chain waits on memory (profiles/r05_ablation_resident_stack.txt).  Here ONE
// FoV supplies a second chain per SIMD itself: workgroups of 80 voxels, two per
// CU, the two on a CU taken from halves of the FoV that are not neighbours, and
// PACED (ConvStackTab::pace) so that one's taps fall into the other's wait /
// stage / drain / publish, layer after layer (profiles/r06_gate_two_chains.txt:
// free-running, the two chains of a CU drift through each other and gain nothing).
//
// A workgroup = 80 consecutive dense voxels, four waves:
//   * wave w owns the 16-position tile w (voxels 16 w ..) for all 27 taps, on
//     v_mfma_f32_16x16x32_f16 (the flops per clock of 32x32x16);
//   * the FIFTH tile (voxels 64 .. 79) is split over the waves by tap: wave w
//     takes taps w, w + 4, ... of it, with the weight fragments it holds for its
//     own tile at that tap anyway; the four partial sums meet in LDS behind the
//     loop and wave 0 finishes the tile.  (450 workgroups cover the 33^3 FoV,
//     two per CU all resident at once; 64-voxel workgroups would need 562 slots.)
//   * the weights go through an LDS ring of TWO units of four taps (16 KB each,
//     four 1-KiB DMA pieces per wave and unit): one barrier per UNIT, and every
//     wave has exactly one double tap (own + fifth tile) per unit;
//   * activations: dz = -1 and dz = 0 segments of 192 rows, dz = +1 into dz = -1's
//     slot once every wave is past tap 6's reads (queued at tap 7): 2 x 24 KB + 32 KB
//     = exactly 80 KB.
//
// Arithmetic: x ~= hi + 2^-11 res as everywhere in this family; per tap and
// 16-channel half of the outputs h:  accC[h] += Whi[h] Xres;  acc[h] += Whi[h] Xhi;
// accC[h] += Wres[h] Xhi  (K = 32: ALL input channels in one instruction), taps
// in order, out = acc + 2^-11 accC; the fifth tile: per-wave partial sums over the
// wave's taps, added in wave order.  Another summation order than conv32m's: the
// same tolerance against the oracle, not the same bits.
//
// Fragments (16x16x32): A = weights, lane l: out channel 16 h + (l & 15), input
// channels 8 (l >> 4) .. + 7;  B = activations, lane l: position l & 15, the same
// input channels = split plane l >> 4 (hi) / 4 + (l >> 4) (residual): the split
// planes of conv32m feed it unchanged;  D: lane l holds position l & 15, out
// channels 16 h + 4 (l >> 4) .. + 3.
// ---------------------------------------------------------------------------
constexpr int kHChunk = 80;
constexpr int kHRows = 192;  // 80 voxels + 3 row ends + one plane end (XS) + 2 (XS + 1), XS <= 34
constexpr int kHSeg = 8 * kHRows * 16;            // bytes of a segment slot (24 DMA pieces)
constexpr int kHPieces = kHSeg / 1024 / 4;        // per wave and segment: 6
constexpr int kHRingOff = 2 * kHSeg;
constexpr int kHUnit = 4 * 4096;                  // four taps of weight fragments
constexpr int kHLdsBytes = kHRingOff + 2 * kHUnit;  // 81,920: two per CU
static_assert(kHSeg % 4096 == 0 && kHLdsBytes <= 80 * 1024, "two workgroups per CU");
// (timing-only experiment builds: treat the FoV as its first FFN_H_VCLIP voxels)
#ifndef FFN_H_VCLIP
#define FFN_H_VCLIP 0
#endif

template <int KIND, bool ADD_SKIP, bool HEAD, bool FLOW, bool RES>
__device__ __forceinline__ void conv32h_body(const ConvDArgs& a, const ConvLayer& L,
                                             const int item, const int v0, const int gc,
                                             f32x4* xres = nullptr) {
  typedef f16x8 frag_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int R = kHRows;
  constexpr int R16 = R * 16;
  constexpr int P = kHPieces;
  constexpr bool kSkipLoad = ADD_SKIP && !RES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
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
  const int li = lane & 15;
  const int lg = lane >> 4;
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsb;

  // ---- weight ring: unit u (taps 4 u .. 4 u + 3) -> slot u & 1; wave w copies fragment
  // w = 2 h + (hi, res) of each of its four taps ----
  auto dma_unit = [&](int u) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      lds_dma16<false, FLOW>(L.wpack + (long)(4 * u + t) * kDTapBytes + wave * 1024,
                             (unsigned)lane * 16,
                             lbase + kHRingOff + (u & 1) * kHUnit + t * 4096 + wave * 1024);
  };
  dma_unit(0);
  dma_unit(1);
  // ---- activations: dz = -1 -> slot 0, dz = 0 -> slot 1, dz = +1 -> slot 0 (tap 7) ----
  const char* g0 = L.in_sp + (long)item * a.item_bytes + (long)p_lo * 16;
  unsigned voff[P];
#pragma unroll
  for (int k = 0; k < P; ++k) {
    const int u = 64 * (wave + 4 * k) + lane;
    const int cp = u / R;
    voff[k] = (unsigned)(cp * (int)a.sp_plane_bytes + (u - cp * R) * 16);
  }
  auto dma_seg = [&](int seg) {  // seg 0, 1, 2 = dz -1, 0, +1
#pragma unroll
    for (int k = 0; k < P; ++k)
      lds_dma16<FLOW, FLOW>(g0 + (long)(seg - 1) * a.plane * 16, voff[k],
                            lbase + (seg & 1) * kHSeg + 64 * (wave + 4 * k) * 16);
  };
  if constexpr (FLOW) {
    if (L.flow_wait_on && !(kAbl & 128)) {
      if (wave == 0) {
        int d_hi = v0 + kHChunk - 1 + a.flow_halo;
        if (FFN_H_VCLIP && d_hi > FFN_H_VCLIP - 1) d_hi = FFN_H_VCLIP - 1;
        flow_wait_tiles(a, L, v0 - a.flow_halo, d_hi, lane);
      }
      ft[1] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  dma_seg(0);
  dma_seg(1);

  // this lane's positions: its own tile (= its wave) and the fifth tile
  const int jpos = wave * 16 + li;
  const bool ok = v0 + jpos < a.V;
  const int ppos = padded(v0 + jpos);
  const int xb = (ppos - p_lo) * 16 + lg * R16;
  const int jpos5 = 64 + li;
  const bool ok5 = v0 + jpos5 < a.V;
  const int ppos5 = padded(v0 + jpos5);
  const int xb5 = (ppos5 - p_lo) * 16 + lg * R16;

  struct XFrag { frag_t hi, res; };
  struct WFrag { frag_t w[2][2]; };  // [out half h][hi, res]
  auto tap_off = [&](int s) {  // LDS byte offset of tap s inside the image (s <= 26)
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    return (kz & 1) * kHSeg + ((ky - 1) * a.XS + (kx - 1)) * 16;
  };
  auto load_x = [&](int s, int base, XFrag& f) {
    const char* px = ldsb + base + tap_off(s);
    f.hi = *reinterpret_cast<const frag_t*>(px);
    f.res = *reinterpret_cast<const frag_t*>(px + 4 * R16);
  };
  auto load_w = [&](int s, int h, WFrag& f) {
    const char* pw = ldsb + kHRingOff + ((s >> 2) & 1) * kHUnit + (s & 3) * 4096 + lane * 16;
    f.w[h][0] = *reinterpret_cast<const frag_t*>(pw + (h * 2 + 0) * 1024);
    f.w[h][1] = *reinterpret_cast<const frag_t*>(pw + (h * 2 + 1) * 1024);
  };
  f32x4 acc[2], accC[2], acc5[2], accC5[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[h][r] = accC[h][r] = acc5[h][r] = accC5[h][r] = 0.f;
  auto mma = [](const frag_t& fw, const frag_t& fx, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, fx, c, 0, 0, 0);
  };
  f32x4 bias4[2], skip4[2], hw4[2], skip5[2];
  float seedv = 0.f, seedv5 = 0.f;

  XFrag X[3], X5;
  WFrag W[2];
  // queue so far: unit 0, unit 1, dz = -1, dz = 0
  wait_vmcnt<P>();  // units 0 and 1, dz = -1 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (FLOW) ft[2] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  load_w(0, 0, W[0]);
  load_w(0, 1, W[0]);
  load_x(0, xb, X[0]);
  load_x(1, xb, X[1]);
  if (wave < 2) load_x(wave, xb5, X5);  // the fifth tile's first tap of waves 0 and 1

  auto issue_epilogue_loads = [&]() {
    const unsigned vb = (unsigned)lg * 16;  // channels 16 h + 4 lg .. + 3
    const char* bp = reinterpret_cast<const char*>(L.bias);
    bias4[0] = hidden_load16f<0, false, FLOW>(bp, vb);
    bias4[1] = hidden_load16f<64, false, FLOW>(bp, vb);
    if constexpr (kSkipLoad) {
      // f32 plane 4 h + lg, 16 B per position; the fifth tile's by every wave (wave 0 uses it)
      const char* xs = reinterpret_cast<const char*>(a.x_f32) + (long)item * a.item_bytes;
      const unsigned vs = (unsigned)(lg * (int)a.sp_plane_bytes + ppos * 16);
      skip4[0] = hidden_load16f<0, FLOW, FLOW>(xs, vs);
      skip4[1] = hidden_load16f<0, FLOW, FLOW>(xs + 4 * a.sp_plane_bytes, vs);
      const unsigned vs5 = (unsigned)(lg * (int)a.sp_plane_bytes + ppos5 * 16);
      skip5[0] = hidden_load16f<0, FLOW, FLOW>(xs, vs5);
      skip5[1] = hidden_load16f<0, FLOW, FLOW>(xs + 4 * a.sp_plane_bytes, vs5);
    }
    if constexpr (HEAD) {
      const char* hp = reinterpret_cast<const char*>(a.head_w);
      hw4[0] = hidden_load16f<0, false, FLOW>(hp, vb);
      hw4[1] = hidden_load16f<64, false, FLOW>(hp, vb);
      const char* sp = reinterpret_cast<const char*>(a.seed_raw + (size_t)item * a.V);
      const unsigned so = (unsigned)(caller_index(a, ok ? v0 + jpos : 0) * 4);
      const unsigned so5 = (unsigned)(caller_index(a, ok5 ? v0 + jpos5 : 0) * 4);
#define FFN_H_SEED(DST, OFF)                                                              \
  if constexpr (FLOW)                                                                     \
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(DST) : "v"(OFF), "s"(sp) : "memory"); \
  else                                                                                    \
    asm volatile("global_load_dword %0, %1, %2" : "=v"(DST) : "v"(OFF), "s"(sp) : "memory");
      FFN_H_SEED(seedv, so)
      FFN_H_SEED(seedv5, so5)
#undef FFN_H_SEED
    }
  };
  // Tap S (unit S >> 2).  At the first tap of the unit's LAST quarter (S & 3 == 3) stands
  // the unit's barrier B: this wave's pieces of unit (S >> 2) + 1 have landed, every wave
  // is past its reads of the unit before -- whose slot unit (S >> 2) + 2 may take now.
  // Between the MFMAs: the weight reads of tap S + 1, the activation reads of tap S + 2
  // (own tile; the fifth tile's for the wave whose double tap S + 2 is).
#define FFN_HGAP(BODY)                       \
  __builtin_amdgcn_sched_barrier(0);         \
  BODY;                                      \
  __builtin_amdgcn_sched_barrier(0);
#define FFN_HTAP(S)                                                                        \
  {                                                                                        \
    WFrag& WCUR = W[(S) & 1];                                                              \
    WFrag& WNEXT = W[((S) + 1) & 1];                                                       \
    XFrag& XCUR = X[(S) % 3];                                                              \
    XFrag& XNEXT = X[((S) + 2) % 3];                                                       \
    if (((S) & 3) == 3 && (S) < 27) {                                                      \
      /* queue behind unit (S>>2)+1: S = 11: the dz = +1 pieces (queued at tap 7) */       \
      if ((S) >= 7) { if ((S) == 11) wait_vmcnt<P>(); else wait_vmcnt<0>(); }              \
      __builtin_amdgcn_s_barrier();                                                        \
      asm volatile("" ::: "memory");                                                       \
      if (((S) >> 2) + 2 <= 6 && !(kAbl & 4)) dma_unit(((S) >> 2) + 2);                    \
      if ((S) == 7) dma_seg(2);                                                            \
      if ((S) == 23) issue_epilogue_loads();                                               \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    accC[0] = mma(WCUR.w[0][0], XCUR.res, accC[0]);                                        \
    FFN_HGAP(if ((S) + 1 <= 26) load_w((S) + 1, 0, WNEXT))                                 \
    accC[1] = mma(WCUR.w[1][0], XCUR.res, accC[1]);                                        \
    FFN_HGAP(if ((S) + 1 <= 26) load_w((S) + 1, 1, WNEXT))                                 \
    acc[0] = mma(WCUR.w[0][0], XCUR.hi, acc[0]);                                           \
    FFN_HGAP(if ((S) + 2 <= 26) load_x((S) + 2, xb, XNEXT))                                \
    acc[1] = mma(WCUR.w[1][0], XCUR.hi, acc[1]);                                           \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    accC[0] = mma(WCUR.w[0][1], XCUR.hi, accC[0]);                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    accC[1] = mma(WCUR.w[1][1], XCUR.hi, accC[1]);                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (wave == ((S) & 3)) {                                                               \
      /* this wave's share of the fifth tile: tap S with the fragments at hand */          \
      accC5[0] = mma(WCUR.w[0][0], X5.res, accC5[0]);                                      \
      accC5[1] = mma(WCUR.w[1][0], X5.res, accC5[1]);                                      \
      acc5[0] = mma(WCUR.w[0][0], X5.hi, acc5[0]);                                         \
      acc5[1] = mma(WCUR.w[1][0], X5.hi, acc5[1]);                                         \
      accC5[0] = mma(WCUR.w[0][1], X5.hi, accC5[0]);                                       \
      accC5[1] = mma(WCUR.w[1][1], X5.hi, accC5[1]);                                       \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if ((S) + 2 <= 26 && wave == (((S) + 2) & 3)) load_x((S) + 2, xb5, X5);                \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
  FFN_HTAP(0) FFN_HTAP(1) FFN_HTAP(2) FFN_HTAP(3) FFN_HTAP(4) FFN_HTAP(5) FFN_HTAP(6)
  FFN_HTAP(7) FFN_HTAP(8) FFN_HTAP(9) FFN_HTAP(10) FFN_HTAP(11) FFN_HTAP(12) FFN_HTAP(13)
  FFN_HTAP(14) FFN_HTAP(15) FFN_HTAP(16) FFN_HTAP(17) FFN_HTAP(18) FFN_HTAP(19) FFN_HTAP(20)
  FFN_HTAP(21) FFN_HTAP(22) FFN_HTAP(23) FFN_HTAP(24) FFN_HTAP(25) FFN_HTAP(26)
#undef FFN_HTAP
#undef FFN_HGAP
  if constexpr (FLOW) ft[3] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;

  // ---- the fifth tile's partial sums meet in LDS (slot 1: the dz = 0 rows are dead
  // since tap 17): P5[wave][h][lane] x 16 B ----
  wait_vmcnt<0>();
  asm volatile("" : "+v"(bias4[0]), "+v"(bias4[1]));
  if constexpr (kSkipLoad)
    asm volatile("" : "+v"(skip4[0]), "+v"(skip4[1]), "+v"(skip5[0]), "+v"(skip5[1]));
  if constexpr (HEAD) asm volatile("" : "+v"(hw4[0]), "+v"(hw4[1]), "+v"(seedv), "+v"(seedv5));
  {
    char* p5 = ldsb + kHSeg + (wave * 2) * 1024 + lane * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      *reinterpret_cast<f32x4*>(p5 + h * 1024) = acc5[h] + accC5[h] * 4.8828125e-4f;  // 2^-11
  }
  __syncthreads();

  // ---- epilogue: straight from the accumulators (lane = position, register i of half h =
  // channel 16 h + 4 lg + i); wave 0 then the fifth tile from the four partial sums ----
  if constexpr (ADD_SKIP && RES) {
    skip4[0] = xres[0];
    skip4[1] = xres[1];
    skip5[0] = xres[2];
    skip5[1] = xres[3];
  }
  unsigned range_max = 0;
  unsigned mine = 0;
  const __amdgpu_buffer_rsrc_t rs_sp = __builtin_amdgcn_make_buffer_rsrc(
      L.out_sp + (long)item * a.item_bytes, 0, a.sp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.x_f32) + (long)item * a.item_bytes, 0, a.sp_bytes, 0x00020000);
  // one 16-position tile: sums s[2] (half h), residual input sk[2], residual output xr[2]
  auto finish_tile = [&](const f32x4 (&s)[2], const f32x4 (&sk)[2], f32x4* xr, const bool okv,
                         const int pp, const int jp, const float sdv) {
    if constexpr (HEAD) {
      const float hbias = a.head_w[kFeatures];
      float partial = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 v = s[h] + bias4[h];
        if (ADD_SKIP) v += sk[h];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          partial = __builtin_fmaf(fmaxf(v[i], 0.f), hw4[h][i], partial);
      }
      partial += __shfl_xor(partial, 16);  // the other channel quads of the position
      partial += __shfl_xor(partial, 32);
      bool above = false;
      if (lg == 0 && okv) {
        const size_t dv = (size_t)item * a.V + caller_index(a, v0 + jp);
        float sd = sdv;
        if (sd != sd) sd = a.pad_value;
        const float lgt = sd + (partial + hbias);
        a.logits[dv] = lgt;
        above = lgt >= a.move_thr;
      }
      mine += (unsigned)__popcll(__ballot(above));
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 v = s[h] + bias4[h];
        if (KIND == 1) {
          if (ADD_SKIP) v += sk[h];
          if constexpr (RES) {
            xr[h] = v;
          } else {
            const unsigned xo =
                okv ? (unsigned)((4 * h + lg) * (int)a.sp_plane_bytes + pp * 16) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x, xo, 0, 16);
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
        // channels 16 h + 4 lg ..: split plane 2 h + (lg >> 1), half lg & 1 of its 16 B
        const unsigned so =
            okv ? (unsigned)((2 * h + (lg >> 1)) * (int)a.sp_plane_bytes + pp * 16 + (lg & 1) * 8)
                : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rs_sp, so, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r4), rs_sp, so,
                                              (int)(4 * a.sp_plane_bytes), 16);
      }
    }
  };
  {
    f32x4 s[2];
    s[0] = acc[0] + accC[0] * 4.8828125e-4f;  // 2^-11
    s[1] = acc[1] + accC[1] * 4.8828125e-4f;
    finish_tile(s, skip4, xres, ok, ppos, jpos, seedv);
  }
  if (wave == 0) {
    f32x4 s[2];
    const char* p5 = ldsb + kHSeg + lane * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s[h] = *reinterpret_cast<const f32x4*>(p5 + h * 1024);
#pragma unroll
      for (int w = 1; w < 4; ++w)
        s[h] += *reinterpret_cast<const f32x4*>(p5 + (w * 2 + h) * 1024);
    }
    finish_tile(s, skip5, xres ? xres + 2 : nullptr, ok5, ppos5, jpos5, seedv5);
  }
  if constexpr (HEAD) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* cnt = reinterpret_cast<float*>(ldsb);
    if (lane == 0) cnt[wave] = __uint_as_float(mine);
    __syncthreads();
    if (tid == 0)
      a.head_count[gc] = __float_as_uint(cnt[0]) + __float_as_uint(cnt[1]) +
                         __float_as_uint(cnt[2]) + __float_as_uint(cnt[3]);
  } else {
    if (!(FLOW && kAbl) && __ballot(range_max > 0x477fe000u) && lane == 0)  // > 65504 (or NaN)
      *a.range_flag = a.range_tag;
    if constexpr (FLOW) ft[4] = flow_publish(a, L, v0, tid);
  }
  if constexpr (FLOW) {
    ft[5] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
    flow_trace_row(a, L, gc, ft);
  }
}

// workgroup -> chunk: the first `per_first` blocks of an XCD take the CUs' first
// slots, the next `per_second` their second ones (dispatch order; speed only): the
// two workgroups of a CU get chunks half a FoV apart
struct ConvHalfMap {
  int n_chunks;    // 80-voxel chunks of the FoV
  int n_first;     // chunks [0, n_first) on the first slots, the rest on the second
  int per_first;   // blocks per XCD on first slots (= its CUs: the dispatcher fills those first)
  int per_second;  // ... and on second slots
};

__device__ __forceinline__ int half_chunk(const ConvHalfMap& mp) {
  const int xcd = blockIdx.x & 7;
  const int i = blockIdx.x >> 3;
  const bool first = i < mp.per_first;
  const int c = first ? xcd * mp.per_first + i : mp.n_first + xcd * mp.per_second + (i - mp.per_first);
  return c < (first ? mp.n_first : mp.n_chunks) ? c : -1;
}

// one conv as its own launch (several FoVs never come here: conv32m takes them; this
// is the repeat of a voided resident step, and flow = 0): the same bits as conv32hs
template <int KIND, bool ADD_SKIP, bool HEAD>
__global__ __launch_bounds__(kDThreads, 2) void conv32h_kernel(ConvDArgs a, ConvHalfMap mp) {
  const int c = half_chunk(mp);
  if (c < 0) return;
  conv32h_body<KIND, ADD_SKIP, HEAD, false, false>(a, a.L, 0, c * kHChunk, c);
}

// the whole stack of ONE FoV, resident (conv32ps's loop over conv32h's bodies)
__global__ __launch_bounds__(kDThreads, 2) void conv32hs_kernel(ConvDArgs a, ConvHalfMap mp,
                                                                ConvStackTab tb) {
  const int c = half_chunk(mp);
  if (c < 0) return;
  const int v0 = c * kHChunk;
  // the residual stream of the workgroup's voxels stays in registers: [0, 1] this wave's
  // tile, [2, 3] the fifth tile (used by wave 0)
  f32x4 xres[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xres[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* wp = tb.wpack0 + (long)tb.l_begin * tb.wpack_stride;
  const float* bp = tb.bias0 + (long)tb.l_begin * tb.bias_stride;
  const long wstride = tb.wpack_stride, bstride = tb.bias_stride;
  unsigned epoch = tb.epoch0 + (unsigned)tb.l_begin;
  const char* sp_in = (tb.l_begin & 1) ? tb.sp_s : tb.sp_t;
  const char* sp_out = (tb.l_begin & 1) ? tb.sp_t : tb.sp_s;
  const int l_first = tb.l_begin;
  // Pacing (tb.pace > 0): conv l of chunk c does not start before  t0 + l pace + phi(c),
  // phi(c) = pace_spread c / n_chunks -- with pace_spread = pace the two workgroups of a CU
  // (chunks about n / 2 apart) sit half a period apart, neighbours in the FoV within a few percent of one: a workgroup's
  // taps then fall into its CU-mate's wait / stage / drain, layer after layer, instead
  // of wherever the free-running hand-off leaves them.  Timing only.
  const long long t_pace0 = wall_clock64() + (long long)tb.pace_spread * c / mp.n_chunks;
  for (int l = tb.l_begin; l < tb.l_end; ++l) {
    if (tb.pace > 0) {
      const long long target = t_pace0 + (long long)(l - l_first) * tb.pace;
      while (wall_clock64() < target) __builtin_amdgcn_s_sleep(1);
    }
    ConvLayer L;
    L.in_sp = sp_in;
    L.out_sp = const_cast<char*>(sp_out);
    L.wpack = wp;
    L.bias = bp;
    L.dbg = nullptr;
    L.flow_wait = epoch;
    L.flow_set = epoch + 1u;
    L.flow_wait_on = l > l_first;
    L.layer = l;
    wp += wstride;
    bp += bstride;
    epoch += 1u;
    {
      const char* t = sp_in;
      sp_in = sp_out;
      sp_out = t;
    }
    const bool last = l == tb.nlayers - 1;
    if (l == 0)
      conv32h_body<1, false, false, true, true>(a, L, 0, v0, c, xres);
    else if (last)
      conv32h_body<1, true, true, true, true>(a, L, 0, v0, c, xres);
    else if (l & 1)
      conv32h_body<0, false, false, true, true>(a, L, 0, v0, c, xres);
    else
      conv32h_body<1, true, false, true, true>(a, L, 0, v0, c, xres);
  }
}

}  // namespace ffn
