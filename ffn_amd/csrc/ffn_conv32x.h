// conv32x (conv_variant 9, a step of ONE FoV): conv32m with the FoV dealt EVENLY
// over the CUs -- no tail workgroups.
//
// conv32mt gives 256 CUs a 128-voxel conv32m workgroup each and the voxels past
// them (33^3: 3,169) to 100 K-split tail workgroups on 100 of those CUs.  A tail
// workgroup pulls 108 KB of weight fragments + 55 KB of activations through its
// CU's 64 B/clk vector-memory path on top of the main workgroup's 198 KB, and
// adds 42 MFMAs to each SIMD: the launch ends with those CUs (7.0 - 7.5 us
// against 6.0 for a CU that hosts a main workgroup alone,
// profiles/r02_wg_timeline.txt).  Here every workgroup takes ceil(V / 256)
// voxels (33^3: 255 workgroups of 141): its four waves keep ONE 32-position tile
// each for all 27 taps, exactly as in conv32m, and share a fifth, partial tile
// (the chunk's voxels past the 128th): the fifth tile uses the weight fragments
// a wave already holds in registers (no extra weight traffic at all) and 16 more
// staged rows per segment (256 instead of 240).
//
// The fifth tile's six products per tap are dealt over the waves so that every
// tap costs every wave the same: waves 0 / 1 take the first K half (kh = 0),
// waves 2 / 3 the second; inside a pair the roles alternate with the tap,
//     "two": M5 += w_hi . x5_hi,  C5 += w_hi . x5_res
//     "one": C5 += w_res . x5_hi
// i.e. one or two extra MFMAs per wave and tap on top of its own six.  The pair
// (kh) is picked by DATA (a v_cndmask on one weight fragment, an address offset
// for the activations), the role by the PROGRAM: a wave runs one of two
// straight-line instantiations (ROLE = wave & 1) chosen once at kernel entry --
// no branch inside the tap loop.  Measured alternatives (profiles/
// r03_ab_conv32x.txt): four per-wave programs (no select at all, DMA issue
// staggered by wave) starve on instruction fetch -- four 8.5-KB streams per CU,
// each executed once: WG duration 7.7 us against 6.8; ONE program with every
// difference as data (eight MFMAs per tap for every wave, one of them times
// zero, 16 v_cndmask per tap in front of them) runs at 60 cycles per MFMA; so
// did the first cut of round 2 (six branches per tap).
// The dz = +1 segment, which conv32m queues in one tap (32 KB: 1.1 K cycles of
// its loop, profiles/r03_ab_conv32m_loop_ablations.txt), is spread over taps
// 9 .. 12.  The four partial sums of
// the fifth tile meet in LDS after the last tap (16 KB, in the dz = 0 slot that
// nobody reads after tap 17) and wave w finishes channels 8 w .. 8 w + 7.
//
// 256-row segments + the ring = 86,016 B of LDS, ~200 registers: ONE workgroup
// per CU, which is what a single FoV gives anyway.  Steps of several FoVs keep
// conv32m (two workgroups per CU).
#pragma once

namespace ffn {

constexpr int kXRows = 256;
constexpr int kXPieces = 8;                    // 8 planes x 256 rows = 8 x 4 x 64 units
constexpr int kXSeg = 8 * kXRows * 16;         // 32,768 B per segment slot
constexpr int kXRing = 2 * kXSeg;
constexpr int kXRingTaps = 5;
constexpr int kXLdsBytes = kXRing + kXRingTaps * 4096;  // 86,016
static_assert(kXPieces == kMPieces, "m_wait counts kMPieces per segment");

// The workgroup computes the a.xch dense voxels from v0 of FoV `item` (the first
// 128 as four tiles, the rest as the shared fifth tile); gc = its slot in
// head_count.
// vmcnt for tap S's wait (cf. m_wait): operations this wave issued before it
// that are NEWER than its piece of W(S+1).  Per tap t, in this order: the ring
// piece W(t+D-1) [t <= 27 - D], two pieces of dz = +1 [t = 9 .. 12], the NEPI
// epilogue operands [t = 27 - D].
constexpr int x_wait(int S, int D, int NEPI) {
  if (S == 0) return kXPieces;      // dz = 0's DMAs are newer than dz = -1 / W1
  if (S + 1 > 26) return -1;
  if (S + 1 <= D - 2) return -1;    // queued in front of everything: landed
  const int tr = S + 2 - D;         // the tap that queued W(S+1)
  int n = 0;
  for (int t = tr; t <= S - 1; ++t) {
    if (t > tr && t <= 27 - D) n += 1;
    if (t >= 9 && t <= 12) n += 2;
    if (t == 27 - D) n += NEPI;
  }
  return n;
}

template <int KIND, bool ADD_SKIP, bool HEAD, int ROLE>
__device__ __forceinline__ void conv32x_body(const ConvDArgs& a, const int item,
                                             const int v0, const int gc,
                                             const bool dbg_here) {
  typedef f16x8 frag_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int R = kXRows;
  constexpr int R16 = R * 16;
  // epilogue operands queued at tap 27 - D: bias 4, [skip 4 + 1], [head 4 + 2]
  constexpr int NEPI = HEAD ? (ADD_SKIP ? 15 : 10) : (ADD_SKIP ? 9 : 4);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const long long dbg_c0 = a.dbg ? clock64() : 0;
  const long long dbg_w0 = a.dbg ? wall_clock64() : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool kh5 = wave >= 2;    // this wave's K half of the fifth tile
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
  constexpr int D = kXRingTaps;
  auto dma_w = [&](int s) {
    lds_dma16(a.wpack + (long)s * kDTapBytes + wave * 1024, (unsigned)lane * 16,
              lbase + kXRing + (s % D) * 4096 + wave * 1024);
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) dma_w(s);
  // ---- activations: dz = -1 -> slot 0, dz = 0 -> slot 1 (dz = +1 later -> slot 0)
  const char* g0 = a.in_sp + (long)item * a.item_bytes + (long)p_lo * 16;
  unsigned voff[kXPieces];
#pragma unroll
  for (int k = 0; k < kXPieces; ++k) {
    const int u = 64 * (wave + 4 * k) + lane;  // unit of 16 B: (plane u / R, row u % R)
    const int cp = u / R;
    voff[k] = (unsigned)(cp * (int)a.sp_plane_bytes + (u - cp * R) * 16);
  }
  auto dma_seg = [&](int seg) {  // seg 0, 1, 2 = dz -1, 0, +1
#pragma unroll
    for (int k = 0; k < kXPieces; ++k)
      lds_dma16(g0 + (long)(seg - 1) * a.plane * 16, voff[k],
                lbase + (seg & 1) * kXSeg + 64 * (wave + 4 * k) * 16);
  };
  dma_seg(0);
  dma_seg(1);

  // this lane's position of its wave's own tile, and of the shared fifth tile
  const int jpos = wave * 32 + li;
  const bool ok = v0 + jpos < a.V;
  const int ppos = padded(v0 + jpos);
  const int xb = (ppos - p_lo) * 16 + lh * R16;
  const int n5 = a.xch - 128;  // positions of the fifth tile (1 .. 32)
  const int j5 = 128 + (li < n5 ? li : n5 - 1);  // (spare lanes repeat the last one)
  const bool ok5 = li < n5 && v0 + j5 < a.V;
  const int ppos5 = padded(v0 + j5);
  const int xb5 = (ppos5 - p_lo) * 16 + lh * R16 + (kh5 ? 2 * R16 : 0);

  struct XFrag { frag_t x[2][2]; };
  struct WFrag { frag_t w[2][2]; };
  struct X5Frag { frag_t hi, res; };
  auto load_x = [&](int s, int kh, XFrag& f) {  // 2 of the 4 activation reads of tap s
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    const char* px = ldsb + xb + (kz & 1) * kXSeg + ((ky - 1) * a.XS + (kx - 1)) * 16;
    f.x[kh][0] = *reinterpret_cast<const frag_t*>(px + (0 * 4 + kh * 2) * R16);
    f.x[kh][1] = *reinterpret_cast<const frag_t*>(px + (1 * 4 + kh * 2) * R16);
  };
  // the fifth tile's fragments of tap s for this wave's K half: hi always, the
  // residual only in a tap where this wave plays "two"
  auto two_at = [](int s) { return ((s + ROLE) & 1) == 0; };
  auto load_x5 = [&](int s, X5Frag& f) {
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    const char* px = ldsb + xb5 + (kz & 1) * kXSeg + ((ky - 1) * a.XS + (kx - 1)) * 16;
    f.hi = *reinterpret_cast<const frag_t*>(px);
    if (two_at(s)) f.res = *reinterpret_cast<const frag_t*>(px + 4 * R16);
  };
  auto load_w = [&](int s, int kh, WFrag& f) {  // 2 of the 4 weight reads of tap s
    const char* pw = ldsb + kXRing + (s % D) * 4096 + lane * 16;
    f.w[kh][0] = *reinterpret_cast<const frag_t*>(pw + (kh * 2 + 0) * 1024);
    f.w[kh][1] = *reinterpret_cast<const frag_t*>(pw + (kh * 2 + 1) * 1024);
  };
  f32x16 acc, accC, acc5, acc5C;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = accC[r] = acc5[r] = acc5C[r] = 0.f;
  auto mma = [](const frag_t& fw, const frag_t& fx, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, c, 0, 0, 0);
  };
  // epilogue operands (hidden loads, issued at tap 27 - D)
  f32x4 bias4[4], skip4[4], hw4[4], skip5;
  float seedv = 0.f, seed5 = 0.f, hbias = 0.f;

  XFrag X0, X1, X2;
  WFrag W0, W1;
  X5Frag Y0, Y1;
  wait_vmcnt<kXPieces>();  // W0 .. W(D-2), dz = -1 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const long long dbg_c1 = a.dbg ? clock64() : 0;
  dma_w(D - 1);
  load_w(0, 0, W0);
  load_w(0, 1, W0);
  load_x(0, 0, X0);
  load_x(0, 1, X0);
  load_x5(0, Y0);
  load_x(1, 0, X1);
  load_x(1, 1, X1);

  auto dma_seg_part = [&](int k0, int k1) {  // pieces [k0, k1) of dz = +1 -> slot 0
#pragma unroll
    for (int k = k0; k < k1; ++k)
      lds_dma16(g0 + (long)a.plane * 16, voff[k], lbase + 64 * (wave + 4 * k) * 16);
  };
  auto issue_epilogue_loads = [&]() {
    const unsigned vb = (unsigned)lh * 16;  // channels 8 g + 4 lh .. + 3
    const char* bp = reinterpret_cast<const char*>(a.bias);
    bias4[0] = hidden_load16f<0>(bp, vb);
    bias4[1] = hidden_load16f<32>(bp, vb);
    bias4[2] = hidden_load16f<64>(bp, vb);
    bias4[3] = hidden_load16f<96>(bp, vb);
    if constexpr (ADD_SKIP) {
      // f32 plane 2 g + lh, 16 B per position
      const char* xs = reinterpret_cast<const char*>(a.x_f32) + (long)item * a.item_bytes;
      const unsigned vs = (unsigned)(lh * (int)a.sp_plane_bytes + ppos * 16);
      skip4[0] = hidden_load16f<0>(xs, vs);
      skip4[1] = hidden_load16f<0>(xs + 2 * a.sp_plane_bytes, vs);
      skip4[2] = hidden_load16f<0>(xs + 4 * a.sp_plane_bytes, vs);
      skip4[3] = hidden_load16f<0>(xs + 6 * a.sp_plane_bytes, vs);
      // the fifth tile: this wave finishes channel group g = wave
      skip5 = hidden_load16f<0>(
          xs + (long)(2 * wave) * a.sp_plane_bytes,
          (unsigned)(lh * (int)a.sp_plane_bytes + ppos5 * 16));
    }
    if constexpr (HEAD) {
      const char* hp = reinterpret_cast<const char*>(a.head_w);
      hw4[0] = hidden_load16f<0>(hp, vb);
      hw4[1] = hidden_load16f<32>(hp, vb);
      hw4[2] = hidden_load16f<64>(hp, vb);
      hw4[3] = hidden_load16f<96>(hp, vb);
      const char* sp = reinterpret_cast<const char*>(a.seed_raw + (size_t)item * a.V);
      asm volatile("global_load_dword %0, %1, %2"
                   : "=v"(seedv)
                   : "v"((unsigned)(caller_index(a, ok ? v0 + jpos : 0) * 4)), "s"(sp)
                   : "memory");
      asm volatile("global_load_dword %0, %1, %2"
                   : "=v"(seed5)
                   : "v"((unsigned)(caller_index(a, ok5 ? v0 + j5 : 0) * 4)), "s"(sp)
                   : "memory");
    }
  };
  // tap S: XCUR / WCUR / YCUR hold its fragments; WNEXT takes tap S + 1's
  // weights, XNEXT tap S + 2's activations, YNEXT tap S + 1's fifth-tile ones.
  // Everything that is not an MFMA sits between the MFMAs (sched_barrier pins
  // the order), in the shadow of the matrix pipe.
#define FFN_XGAP(S, PART, WNEXT, XNEXT)                                          \
  __builtin_amdgcn_sched_barrier(0);                                            \
  if ((PART) < 2 && (S) + 1 <= 26) load_w((S) + 1, PART, WNEXT);                \
  if ((PART) >= 2 && (S) + 2 <= 26) load_x((S) + 2, (PART) - 2, XNEXT);         \
  if ((S) >= 9 && (S) <= 12 && ((PART) == 1 || (PART) == 3))                    \
    dma_seg_part(2 * ((S) - 9) + (PART) / 2, 2 * ((S) - 9) + (PART) / 2 + 1);   \
  __builtin_amdgcn_sched_barrier(0);
#define FFN_XTAP(S, XCUR, WCUR, WNEXT, XNEXT, YCUR, YNEXT)                       \
  {                                                                             \
    if ((S) > 0) {                                                              \
      if constexpr (x_wait(S, D, NEPI) >= 0) wait_vmcnt<x_wait(S, D, NEPI)>();  \
      __builtin_amdgcn_s_barrier();                                             \
      asm volatile("" ::: "memory");                                            \
    }                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                          \
    accC = mma(WCUR.w[0][0], XCUR.x[0][1], accC);                               \
    __builtin_amdgcn_sched_barrier(0);                                          \
    if ((S) > 0 && (S) + D - 1 <= 26) dma_w((S) + D - 1);                       \
    __builtin_amdgcn_sched_barrier(0);                                          \
    acc = mma(WCUR.w[0][0], XCUR.x[0][0], acc);                                 \
    FFN_XGAP(S, 0, WNEXT, XNEXT)                                                \
    accC = mma(WCUR.w[0][1], XCUR.x[0][0], accC);                               \
    FFN_XGAP(S, 1, WNEXT, XNEXT)                                                \
    {                                                                           \
      const frag_t w5 = two_at(S) ? (kh5 ? WCUR.w[1][0] : WCUR.w[0][0])         \
                                  : (kh5 ? WCUR.w[1][1] : WCUR.w[0][1]);        \
      if (two_at(S)) {                                                          \
        acc5 = mma(w5, YCUR.hi, acc5);                                          \
      } else {                                                                  \
        acc5C = mma(w5, YCUR.hi, acc5C);                                        \
      }                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                        \
      acc = mma(WCUR.w[1][0], XCUR.x[1][0], acc);                               \
      FFN_XGAP(S, 2, WNEXT, XNEXT)                                              \
      accC = mma(WCUR.w[1][0], XCUR.x[1][1], accC);                             \
      FFN_XGAP(S, 3, WNEXT, XNEXT)                                              \
      if (two_at(S)) {                                                          \
        acc5C = mma(w5, YCUR.res, acc5C);                                       \
        __builtin_amdgcn_sched_barrier(0);                                      \
      }                                                                         \
    }                                                                           \
    accC = mma(WCUR.w[1][1], XCUR.x[1][0], accC);                               \
    __builtin_amdgcn_sched_barrier(0);                                          \
    if ((S) + 1 <= 26) load_x5((S) + 1, YNEXT);                                 \
    if ((S) == 27 - D) issue_epilogue_loads();                                  \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }
  FFN_XTAP(0, X0, W0, W1, X2, Y0, Y1)
  FFN_XTAP(1, X1, W1, W0, X0, Y1, Y0)
  FFN_XTAP(2, X2, W0, W1, X1, Y0, Y1)
  FFN_XTAP(3, X0, W1, W0, X2, Y1, Y0)
  FFN_XTAP(4, X1, W0, W1, X0, Y0, Y1)
  FFN_XTAP(5, X2, W1, W0, X1, Y1, Y0)
  FFN_XTAP(6, X0, W0, W1, X2, Y0, Y1)
  FFN_XTAP(7, X1, W1, W0, X0, Y1, Y0)
  FFN_XTAP(8, X2, W0, W1, X1, Y0, Y1)
  FFN_XTAP(9, X0, W1, W0, X2, Y1, Y0)
  FFN_XTAP(10, X1, W0, W1, X0, Y0, Y1)
  FFN_XTAP(11, X2, W1, W0, X1, Y1, Y0)
  FFN_XTAP(12, X0, W0, W1, X2, Y0, Y1)
  FFN_XTAP(13, X1, W1, W0, X0, Y1, Y0)
  FFN_XTAP(14, X2, W0, W1, X1, Y0, Y1)
  FFN_XTAP(15, X0, W1, W0, X2, Y1, Y0)
  FFN_XTAP(16, X1, W0, W1, X0, Y0, Y1)
  FFN_XTAP(17, X2, W1, W0, X1, Y1, Y0)
  FFN_XTAP(18, X0, W0, W1, X2, Y0, Y1)
  FFN_XTAP(19, X1, W1, W0, X0, Y1, Y0)
  FFN_XTAP(20, X2, W0, W1, X1, Y0, Y1)
  FFN_XTAP(21, X0, W1, W0, X2, Y1, Y0)
  FFN_XTAP(22, X1, W0, W1, X0, Y0, Y1)
  FFN_XTAP(23, X2, W1, W0, X1, Y1, Y0)
  FFN_XTAP(24, X0, W0, W1, X2, Y0, Y1)
  FFN_XTAP(25, X1, W1, W0, X0, Y1, Y0)
  FFN_XTAP(26, X2, W0, W1, X1, Y0, Y1)
#undef FFN_XTAP
#undef FFN_XGAP
  const long long dbg_c2 = a.dbg ? clock64() : 0;

  // ---- the fifth tile's partial sums -> LDS (the dz = 0 slot: no wave reads it
  // after tap 17, and every wave is past the barrier of tap 18) ----
  {
    const f32x16 s5 = acc5 + acc5C * 4.8828125e-4f;  // 2^-11
    float* red = reinterpret_cast<float*>(ldsb + kXSeg) + wave * (16 * 64) + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[r * 64] = s5[r];
  }

  // ---- epilogue of the wave's own tile: straight from the accumulators (lane =
  // position jpos, register 4 g + i = channel 8 g + 4 lh + i) ----
  wait_vmcnt<0>();
  asm volatile(""
               : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]));
  if constexpr (ADD_SKIP)
    asm volatile(""
                 : "+v"(skip4[0]), "+v"(skip4[1]), "+v"(skip4[2]), "+v"(skip4[3]),
                   "+v"(skip5));
  if constexpr (HEAD)
    asm volatile(""
                 : "+v"(hw4[0]), "+v"(hw4[1]), "+v"(hw4[2]), "+v"(hw4[3]),
                   "+v"(seedv), "+v"(seed5));
  const f32x16 s = acc + accC * 4.8828125e-4f;  // 2^-11
  unsigned range_max = 0;
  const __amdgpu_buffer_rsrc_t rs_sp = __builtin_amdgcn_make_buffer_rsrc(
      a.out_sp + (long)item * a.item_bytes, 0, a.sp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.x_f32) + (long)item * a.item_bytes, 0, a.sp_bytes,
      0x00020000);
  // conv_a / conv_b output of 4 channels (8 g + 4 lh ..) of padded position pp
  auto store_group = [&](f32x4 v, int g, int pp, bool valid) {
    if (KIND == 1) {
      const unsigned xo =
          valid ? (unsigned)((2 * g + lh) * (int)a.sp_plane_bytes + pp * 16)
                : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x, xo,
                                             0, 16);
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
        valid ? (unsigned)(g * (int)a.sp_plane_bytes + pp * 16 + lh * 8)
              : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rs_sp, so, 0,
                                          16);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r4), rs_sp, so,
                                          (int)(4 * a.sp_plane_bytes), 16);
  };
  float head_partial = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
    v += bias4[g];
    if (KIND == 1 && ADD_SKIP) v += skip4[g];
    if constexpr (HEAD) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        head_partial = __builtin_fmaf(fmaxf(v[i], 0.f), hw4[g][i], head_partial);
    } else {
      store_group(v, g, ppos, ok);
    }
  }

  // ---- the fifth tile: the four partial sums meet, wave w finishes channel
  // group g = w (channels 8 w + 4 lh .. + 3 of position li) ----
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  f32x4 v5 = bias4[0];
  {
    // (bias4[g] differs by g only: pick this wave's)
    v5 = wave == 0 ? bias4[0] : wave == 1 ? bias4[1] : wave == 2 ? bias4[2] : bias4[3];
    const float* red = reinterpret_cast<const float*>(ldsb + kXSeg) + lane;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w)  // fixed order: the bits do not depend on timing
#pragma unroll
      for (int i = 0; i < 4; ++i)
        sum[i] += red[w * (16 * 64) + (4 * wave + i) * 64];
    v5 += sum;
    if (KIND == 1 && ADD_SKIP) v5 += skip5;
  }
  if constexpr (HEAD) {
    hbias = a.head_w[kFeatures];
    const f32x4 hw5 =
        wave == 0 ? hw4[0] : wave == 1 ? hw4[1] : wave == 2 ? hw4[2] : hw4[3];
    float p5 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p5 = __builtin_fmaf(fmaxf(v5[i], 0.f), hw5[i], p5);
    p5 += __shfl_xor(p5, 32);            // the other 4 channels of the group
    head_partial += __shfl_xor(head_partial, 32);  // own tile: the other 16
    // per-position partial dot products of the fifth tile: [wave][li] behind the
    // partial sums (which all waves have read: barrier below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* dot = reinterpret_cast<float*>(ldsb + kXSeg);
    if (lh == 0) dot[wave * 32 + li] = p5;
    __syncthreads();
    bool above = false, above5 = false;
    if (lh == 0 && ok) {
      const size_t dv = (size_t)item * a.V + caller_index(a, v0 + jpos);
      float sd = seedv;
      if (sd != sd) sd = a.pad_value;
      const float lg = sd + (head_partial + hbias);
      a.logits[dv] = lg;
      above = lg >= a.move_thr;
    }
    if (wave == 0 && lh == 0 && ok5) {
      const float upd = ((dot[li] + dot[32 + li]) + dot[64 + li]) + dot[96 + li];
      const size_t dv = (size_t)item * a.V + caller_index(a, v0 + j5);
      float sd = seed5;
      if (sd != sd) sd = a.pad_value;
      const float lg = sd + (upd + hbias);
      a.logits[dv] = lg;
      above5 = lg >= a.move_thr;
    }
    const unsigned mine = (unsigned)__popcll(__ballot(above)) +
                          (unsigned)__popcll(__ballot(above5));
    __syncthreads();
    float* cnt = reinterpret_cast<float*>(ldsb + kXSeg + 1024);
    if (lane == 0) cnt[wave] = __uint_as_float(mine);
    __syncthreads();
    if (tid == 0)
      a.head_count[gc] = __float_as_uint(cnt[0]) + __float_as_uint(cnt[1]) +
                         __float_as_uint(cnt[2]) + __float_as_uint(cnt[3]);
  } else {
    store_group(v5, wave, ppos5, ok5);
    if (__ballot(range_max > 0x477fe000u) && lane == 0)  // > 65504 (or NaN)
      *a.range_flag = a.range_tag;
  }
  if (a.dbg && dbg_here && lane == 0) {
    long long* d = a.dbg + wave * 6;
    d[0] = dbg_c0;
    d[1] = dbg_c1;
    d[2] = dbg_c2;
    d[3] = clock64();
    d[4] = dbg_w0;
    d[5] = wall_clock64();
  }
}

template <int KIND, bool ADD_SKIP, bool HEAD>
__global__ __launch_bounds__(kDThreads, 1) void conv32x_kernel(ConvDArgs a) {
  const int gc = (blockIdx.x & 7) * a.slots_per_xcd + (blockIdx.x >> 3);
  if (gc >= a.total_slots) return;
  const long long t0 = a.dbg_wgs ? wall_clock64() : 0;
  const int item = (int)__umulhi((unsigned)gc, a.magic_nchunks);
  const int chunk = gc - item * a.nchunks;
  // the two programs differ in the tap parity at which a wave plays "two"
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1)
    conv32x_body<KIND, ADD_SKIP, HEAD, 1>(a, item, chunk * a.xch, gc, gc == 0);
  else
    conv32x_body<KIND, ADD_SKIP, HEAD, 0>(a, item, chunk * a.xch, gc, gc == 0);
  stamp_workgroup(a, t0);
}

}  // namespace ffn
