// conv32h / conv32hs: the split-product conv on 16-position tiles, 64-voxel workgroups -- two
// independent hand-off chains per SIMD inside ONE FoV (engine option flow = 3).
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

// ---------------------------------------------------------------------------
// Why.  conv32ps (ffn_conv_resident.h) runs one wave per SIMD: a layer's 2.56 us
// of MFMAs sit inside a 7.2-us chain  words seen -> rows staged -> taps ->
// stores drained -> word published  with nothing else to run on the SIMD while
// the chain waits on memory (profiles/r05_ablation_resident_stack.txt).  The same
// convs take 5.2 - 5.4 us per FoV-conv as soon as a second, independent
// workgroup shares the CU (batched steps).  Here ONE FoV supplies that second
// chain itself: workgroups of 64 voxels -- four waves, one 16-position tile each
// on v_mfma_f32_16x16x32_f16 (the same flops per clock as 32x32x16) -- two per
// CU, the two on a CU taken from halves of the FoV that are not neighbours, so
// that one's wait / stage / drain / publish runs under the other's taps.
//
// Arithmetic: x ~= hi + 2^-11 res as everywhere in this family; per tap and
// 16-channel half of the outputs h:  accC[h] += Whi[h] Xres;  acc[h] += Whi[h] Xhi;
// accC[h] += Wres[h] Xhi  (K = 32: ALL input channels in one instruction), taps
// in order, out = acc + 2^-11 accC.  Another K grouping than conv32m's two
// 16-channel halves: same tolerance against the oracle, not the same bits.
//
// Fragments (16x16x32): A = weights, lane l: out channel 16 h + (l & 15), input
// channels 8 (l >> 4) .. + 7;  B = activations, lane l: position l & 15, the same
// input channels = split plane l >> 4 (hi) / 4 + (l >> 4) (residual): the split
// planes of conv32m feed it unchanged;  D: lane l holds position l & 15, out
// channels 16 h + 4 (l >> 4) .. + 3.
// ---------------------------------------------------------------------------
constexpr int kHChunk = 64;
#ifndef FFN_H_ROWS
#define FFN_H_ROWS 176   // 64 voxels + 3 row ends + one plane end (XS) + 2 (XS + 1), XS <= 34
#endif
#ifndef FFN_H_NSEG
#define FFN_H_NSEG 2     // 2: dz = +1 takes dz = -1's slot at tap 9; 3: a slot each
#endif
#ifndef FFN_H_RING
#define FFN_H_RING 8     // taps resident in the weight ring
#endif
constexpr int kHRows = FFN_H_ROWS;
constexpr int kHNSeg = FFN_H_NSEG;
constexpr int kHRing = FFN_H_RING;
constexpr int kHSeg = 8 * kHRows * 16;                       // bytes of a segment slot
constexpr int kHPieces = (kHSeg / 1024 + 3) / 4;             // DMA pieces per wave and segment
constexpr int kHRingOff = kHNSeg * kHSeg;
constexpr int kHLdsBytes = kHRingOff + kHRing * 4096;
static_assert(kHSeg % 1024 == 0, "a segment is whole DMA pieces");
static_assert(kHLdsBytes <= 80 * 1024, "two workgroups per CU");
// (timing-only gate builds: treat the FoV as its first FFN_H_VCLIP voxels)
#ifndef FFN_H_VCLIP
#define FFN_H_VCLIP 0
#endif

// vmcnt for tap S's wait (-1: nothing to wait for): operations issued before it
// that are NEWER than W(S+1).  Issue order: W0 .. W(D-2) | dz=-1 | dz=0 [| dz=+1] |
// tap t: W(t+D-1) [t = 9, two slots: the dz = +1 pieces] [t = 27 - D: NEPI operands]
// (NOW: the tap at whose start the wait stands; S + 1 - NOW = the weight read-ahead)
constexpr int h_wait(int S, int D, int NEPI, int NOW = -1) {
  if (NOW < 0) NOW = S;
  if (S == 0 && NOW == 0) return kHPieces * (kHNSeg - 1);  // dz = -1, the ring's first taps
  if (S + 1 > 26) return -1;
  if (S + 1 <= D - 2) return -1;
  const int tr = S + 2 - D;  // the tap that queued W(S+1)
  int n = 0;
  for (int t = tr; t <= NOW - 1; ++t) {
    if (t > tr && t <= 27 - D) n += 1;
    if (kHNSeg == 2 && t == 9) n += kHPieces;
    if (t == 27 - D) n += NEPI;
  }
  return n;
}

template <int KIND, bool ADD_SKIP, bool HEAD, bool FLOW, bool RES>
__device__ __forceinline__ void conv32h_body(const ConvDArgs& a, const ConvLayer& L,
                                             const int item, const int v0, const int gc,
                                             f32x4* xres = nullptr) {
  typedef f16x8 frag_t;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int R = kHRows;
  constexpr int R16 = R * 16;
  constexpr int D = kHRing;
  constexpr int P = kHPieces;
  constexpr bool kSkipLoad = ADD_SKIP && !RES;
  constexpr int NEPI = HEAD ? (kSkipLoad ? 7 : 5) : (kSkipLoad ? 4 : 2);
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

  // ---- weight ring: tap s -> slot s % D; wave w copies fragment w = 2 h + (hi, res) ----
  auto dma_w = [&](int s) {
    lds_dma16<false, FLOW>(L.wpack + (long)s * kDTapBytes + wave * 1024, (unsigned)lane * 16,
                           lbase + kHRingOff + (s % D) * 4096 + wave * 1024);
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) dma_w(s);
  // ---- activations: segment dz -> slot dz + 1 (two slots: dz = +1 -> slot 0 at tap 9) ----
  const char* g0 = L.in_sp + (long)item * a.item_bytes + (long)p_lo * 16;
  unsigned voff[P];
#pragma unroll
  for (int k = 0; k < P; ++k) {
    int u = 64 * (wave + 4 * k) + lane;
    u = u >= 8 * R ? u - 8 * R : u;
    const int cp = u / R;
    voff[k] = (unsigned)(cp * (int)a.sp_plane_bytes + (u - cp * R) * 16);
  }
  auto dma_seg_part = [&](int seg, int k0, int k1) {  // seg 0, 1, 2 = dz -1, 0, +1
#pragma unroll
    for (int k = k0; k < k1 && k < P; ++k) {
      const int u0 = 64 * (wave + 4 * k);
      lds_dma16<FLOW, FLOW>(g0 + (long)(seg - 1) * a.plane * 16, voff[k],
                            lbase + (kHNSeg == 3 ? seg : (seg & 1)) * kHSeg +
                                (u0 >= 8 * R ? u0 - 8 * R : u0) * 16);
    }
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
  dma_seg_part(0, 0, P);
  dma_seg_part(1, 0, P);
  if constexpr (kHNSeg == 3) dma_seg_part(2, 0, P);

  // this lane's position (its tile = its wave) and its place in the LDS image
  const int jpos = wave * 16 + li;
  const bool ok = v0 + jpos < a.V;
  const int ppos = padded(v0 + jpos);
  const int xb = (ppos - p_lo) * 16 + lg * R16;

  struct XFrag { frag_t hi, res; };
  struct WFrag { frag_t w[2][2]; };  // [out half h][hi, res]
  auto load_x = [&](int s, XFrag& f) {
    const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
    const char* px = ldsb + xb + (kHNSeg == 3 ? kz : (kz & 1)) * kHSeg +
                     ((ky - 1) * a.XS + (kx - 1)) * 16;
    f.hi = *reinterpret_cast<const frag_t*>(px);
    f.res = *reinterpret_cast<const frag_t*>(px + 4 * R16);
  };
  auto load_w = [&](int s, int h, WFrag& f) {
    const char* pw = ldsb + kHRingOff + (s % D) * 4096 + lane * 16;
    f.w[h][0] = *reinterpret_cast<const frag_t*>(pw + (h * 2 + 0) * 1024);
    f.w[h][1] = *reinterpret_cast<const frag_t*>(pw + (h * 2 + 1) * 1024);
  };
  f32x4 acc[2], accC[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[h][r] = accC[h][r] = 0.f;
  auto mma = [](const frag_t& fw, const frag_t& fx, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, fx, c, 0, 0, 0);
  };
  f32x4 bias4[2], skip4[2], hw4[2];
  float seedv = 0.f, hbias = 0.f;

  // fragments in flight: the weights of tap S + WA, the activations of tap S + XA are
  // read during tap S (FFN_H_WAHEAD / FFN_H_XAHEAD; rotating buffers)
#ifndef FFN_H_WAHEAD
#define FFN_H_WAHEAD 2
#endif
#ifndef FFN_H_XAHEAD
#define FFN_H_XAHEAD 3
#endif
  constexpr int WA = FFN_H_WAHEAD, XA = FFN_H_XAHEAD;
  XFrag X[XA + 1];
  WFrag W[WA + 1];
  // W(WA) .. must be in the ring at the first barrier already: D - 1 >= WA + 1
  static_assert(D - 1 >= WA + 1, "ring depth against the weight read-ahead");
  // the activation reads run XA taps ahead of the waits that cover their segments
  static_assert(D < 11 - XA + WA, "tap 9 - XA's wait must stand behind the prologue's DMAs");
  static_assert(kHNSeg == 3 || D <= 9 - XA + WA, "tap 18 - XA's wait must cover tap 9's DMAs");
  wait_vmcnt<h_wait(0, D, NEPI)>();  // W0 .. W(D-2), dz = -1 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (FLOW) ft[2] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;
  dma_w(D - 1);
#pragma unroll
  for (int t = 0; t < WA; ++t) {
    load_w(t, 0, W[t]);
    load_w(t, 1, W[t]);
  }
#pragma unroll
  for (int t = 0; t < XA; ++t) load_x(t, X[t]);

  auto issue_epilogue_loads = [&]() {
    const unsigned vb = (unsigned)lg * 16;  // channels 16 h + 4 lg .. + 3
    const char* bp = reinterpret_cast<const char*>(L.bias);
    bias4[0] = hidden_load16f<0, false, FLOW>(bp, vb);
    bias4[1] = hidden_load16f<64, false, FLOW>(bp, vb);
    if constexpr (kSkipLoad) {
      // f32 plane 4 h + lg, 16 B per position
      const char* xs = reinterpret_cast<const char*>(a.x_f32) + (long)item * a.item_bytes;
      const unsigned vs = (unsigned)(lg * (int)a.sp_plane_bytes + ppos * 16);
      skip4[0] = hidden_load16f<0, FLOW, FLOW>(xs, vs);
      skip4[1] = hidden_load16f<0, FLOW, FLOW>(xs + 4 * a.sp_plane_bytes, vs);
    }
    if constexpr (HEAD) {
      const char* hp = reinterpret_cast<const char*>(a.head_w);
      hw4[0] = hidden_load16f<0, false, FLOW>(hp, vb);
      hw4[1] = hidden_load16f<64, false, FLOW>(hp, vb);
      const char* sp = reinterpret_cast<const char*>(a.seed_raw + (size_t)item * a.V);
      const unsigned so = (unsigned)(caller_index(a, ok ? v0 + jpos : 0) * 4);
      if constexpr (FLOW)
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2"
                     : "=v"(seedv)
                     : "v"(so), "s"(sp)
                     : "memory");
      else
        asm volatile("global_load_dword %0, %1, %2" : "=v"(seedv) : "v"(so), "s"(sp) : "memory");
    }
  };
  // tap S: wait for W(S+WA) in the ring, barrier; six MFMAs with everything else
  // between them: the ring piece of tap S + D - 1 [, the dz = +1 pieces, the epilogue
  // operands], the four weight reads of tap S + WA, the two activation reads of tap S + XA
#define FFN_HGAP(BODY)                       \
  __builtin_amdgcn_sched_barrier(0);         \
  BODY;                                      \
  __builtin_amdgcn_sched_barrier(0);
#define FFN_HTAP(S)                                                                        \
  {                                                                                        \
    WFrag& WCUR = W[(S) % (WA + 1)];                                                       \
    WFrag& WNEXT = W[((S) + WA) % (WA + 1)];                                               \
    XFrag& XCUR = X[(S) % (XA + 1)];                                                       \
    XFrag& XNEXT = X[((S) + XA) % (XA + 1)];                                               \
    if ((S) > 0) {                                                                         \
      if constexpr (h_wait((S) + WA - 1, D, NEPI, (S)) >= 0)                               \
        wait_vmcnt<h_wait((S) + WA - 1, D, NEPI, (S))>();                                  \
      if (!(FLOW && (kAbl & 2))) __builtin_amdgcn_s_barrier();                             \
      asm volatile("" ::: "memory");                                                       \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    accC[0] = mma(WCUR.w[0][0], XCUR.res, accC[0]);                                        \
    FFN_HGAP(if (!(kAbl & 4) && (S) > 0 && (S) + D - 1 <= 26) dma_w((S) + D - 1);                         \
             if (!(kAbl & 8) && (S) + WA <= 26) load_w((S) + WA, 0, WNEXT))                               \
    accC[1] = mma(WCUR.w[1][0], XCUR.res, accC[1]);                                        \
    FFN_HGAP(if (!(kAbl & 8) && (S) + WA <= 26) load_w((S) + WA, 1, WNEXT);                               \
             if (kHNSeg == 2 && (S) == 9) dma_seg_part(2, 0, (P + 2) / 3))                 \
    acc[0] = mma(WCUR.w[0][0], XCUR.hi, acc[0]);                                           \
    FFN_HGAP(if (!(kAbl & 8) && (S) + XA <= 26) load_x((S) + XA, XNEXT);                                  \
             if (kHNSeg == 2 && (S) == 9) dma_seg_part(2, (P + 2) / 3, 2 * ((P + 2) / 3))) \
    acc[1] = mma(WCUR.w[1][0], XCUR.hi, acc[1]);                                           \
    FFN_HGAP(if (kHNSeg == 2 && (S) == 9) dma_seg_part(2, 2 * ((P + 2) / 3), P))           \
    accC[0] = mma(WCUR.w[0][1], XCUR.hi, accC[0]);                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    accC[1] = mma(WCUR.w[1][1], XCUR.hi, accC[1]);                                         \
    FFN_HGAP(if ((S) == 27 - D) issue_epilogue_loads())                                    \
  }
  FFN_HTAP(0) FFN_HTAP(1) FFN_HTAP(2) FFN_HTAP(3) FFN_HTAP(4) FFN_HTAP(5) FFN_HTAP(6)
  FFN_HTAP(7) FFN_HTAP(8) FFN_HTAP(9) FFN_HTAP(10) FFN_HTAP(11) FFN_HTAP(12) FFN_HTAP(13)
  FFN_HTAP(14) FFN_HTAP(15) FFN_HTAP(16) FFN_HTAP(17) FFN_HTAP(18) FFN_HTAP(19) FFN_HTAP(20)
  FFN_HTAP(21) FFN_HTAP(22) FFN_HTAP(23) FFN_HTAP(24) FFN_HTAP(25) FFN_HTAP(26)
#undef FFN_HTAP
#undef FFN_HGAP
  if constexpr (FLOW) ft[3] = (FFN_FLOW_TRACE && a.flow_trace) ? wall_clock64() : 0;

  // ---- epilogue: straight from the accumulators (lane = position jpos, register i of
  // half h = channel 16 h + 4 lg + i) ----
  wait_vmcnt<0>();
  asm volatile("" : "+v"(bias4[0]), "+v"(bias4[1]));
  if constexpr (kSkipLoad) asm volatile("" : "+v"(skip4[0]), "+v"(skip4[1]));
  if constexpr (ADD_SKIP && RES) {
    skip4[0] = xres[0];
    skip4[1] = xres[1];
  }
  if constexpr (HEAD) asm volatile("" : "+v"(hw4[0]), "+v"(hw4[1]), "+v"(seedv));
  unsigned range_max = 0;
  if constexpr (HEAD) {
    hbias = a.head_w[kFeatures];
    float partial = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = acc[h] + accC[h] * 4.8828125e-4f;  // 2^-11
      v += bias4[h];
      if (ADD_SKIP) v += skip4[h];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        partial = __builtin_fmaf(fmaxf(v[i], 0.f), hw4[h][i], partial);
    }
    partial += __shfl_xor(partial, 16);  // the other channel quads of the position
    partial += __shfl_xor(partial, 32);
    bool above = false;
    if (lg == 0 && ok) {
      const size_t dv = (size_t)item * a.V + caller_index(a, v0 + jpos);
      float sd = seedv;
      if (sd != sd) sd = a.pad_value;
      const float lgt = sd + (partial + hbias);
      a.logits[dv] = lgt;
      above = lgt >= a.move_thr;
    }
    const unsigned mine = (unsigned)__popcll(__ballot(above));
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
    for (int h = 0; h < 2; ++h) {
      f32x4 v = acc[h] + accC[h] * 4.8828125e-4f;  // 2^-11
      v += bias4[h];
      if (KIND == 1) {
        if (ADD_SKIP) v += skip4[h];
        if constexpr (RES) {
          xres[h] = v;
        } else {
          const unsigned xo =
              ok ? (unsigned)((4 * h + lg) * (int)a.sp_plane_bytes + ppos * 16) : 0x80000000u;
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
          ok ? (unsigned)((2 * h + (lg >> 1)) * (int)a.sp_plane_bytes + ppos * 16 + (lg & 1) * 8)
             : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rs_sp, so, 0, 16);
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
}

// workgroup -> chunk: the first `per_slot` blocks of an XCD take the CUs' first
// slots, the next `per_slot` their second ones (dispatch order; speed only): the
// two workgroups of a CU get chunks half a FoV apart
struct ConvHalfMap {
  int n_chunks;   // 64-voxel chunks of the FoV
  int n_first;    // chunks [0, n_first) on the first slots, the rest on the second
  int per_slot;   // blocks per XCD and slot
};

// one conv as its own launch (the repeat of a voided resident step; flow = 0)
template <int KIND, bool ADD_SKIP, bool HEAD>
__global__ __launch_bounds__(kDThreads, 2) void conv32h_kernel(ConvDArgs a, ConvHalfMap mp) {
  const int xcd = blockIdx.x & 7;
  const int i = blockIdx.x >> 3;
  const bool first = i < mp.per_slot;
  const int c = first ? xcd * mp.per_slot + i : mp.n_first + xcd * mp.per_slot + (i - mp.per_slot);
  if (c >= (first ? mp.n_first : mp.n_chunks)) return;
  conv32h_body<KIND, ADD_SKIP, HEAD, false, false>(a, a.L, 0, c * kHChunk, c);
}

// the whole stack of ONE FoV, resident (conv32ps's loop over conv32h's bodies)
__global__ __launch_bounds__(kDThreads, 2) void conv32hs_kernel(ConvDArgs a, ConvHalfMap mp,
                                                                ConvStackTab tb) {
  const int xcd = blockIdx.x & 7;
  const int i = blockIdx.x >> 3;
  const bool first = i < mp.per_slot;
  const int c = first ? xcd * mp.per_slot + i : mp.n_first + xcd * mp.per_slot + (i - mp.per_slot);
  if (c >= (first ? mp.n_first : mp.n_chunks)) return;
  const int v0 = c * kHChunk;
  // (experiment, flow_dbg 1024 / 4096: the first / the second slot's waves ahead in
  // the SIMDs' arbitration)
  if (kExp && (a.flow_dbg & 1024) && first) __builtin_amdgcn_s_setprio(1);
  if (kExp && (a.flow_dbg & 4096) && !first) __builtin_amdgcn_s_setprio(1);
  f32x4 xres[2];
  xres[0] = xres[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* wp = tb.wpack0 + (long)tb.l_begin * tb.wpack_stride;
  const float* bp = tb.bias0 + (long)tb.l_begin * tb.bias_stride;
  const long wstride = tb.wpack_stride, bstride = tb.bias_stride;
  unsigned epoch = tb.epoch0 + (unsigned)tb.l_begin;
  const char* sp_in = (tb.l_begin & 1) ? tb.sp_s : tb.sp_t;
  const char* sp_out = (tb.l_begin & 1) ? tb.sp_t : tb.sp_s;
  const int l_first = tb.l_begin;
  // Pacing (tb.pace > 0): conv l of chunk c does not start before  t0 + l pace + phi(c),
  // phi(c) = pace c / n_chunks -- the two workgroups of a CU (chunks c, c + n / 2) half a
  // period apart, neighbours in the FoV within a few percent of one: a workgroup's
  // taps then fall into its CU-mate's wait / stage / drain, layer after layer, instead
  // of wherever the free-running hand-off leaves them.
  const long long t_pace0 = wall_clock64() + (long long)tb.pace * c / mp.n_chunks;
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
