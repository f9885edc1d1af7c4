// conv32ps: the conv stack of a single-FoV step as ONE resident launch (engine option flow = 2).
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

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
  int pace;               // 10-ns ticks between two convs of a workgroup (0: free-running)
  int pace_tail;          // conv32ps: the tail workgroups' offset inside a period
  int pace_spread;        // phi of the FoV's last voxel, in ticks (0: every workgroup at once)
  // [0] when the last step's record was published (faces block), [1] sum over the launches
  // of (this launch's first instruction - [0]) in 10-ns ticks, [2] their count; or NULL
  long long* stamps;
  // A stack queued AHEAD of the host (engine option stack_ahead), behind the speculative
  // conv0_a of its step: that launch's choice word (-1: the device found no valid
  // position and computed nothing -- this launch then ends after its first conv, which
  // ran on the last step's planes; nobody reads what it wrote).  NULL: an ordinary launch.
  const int* ahead_choice;
  int trace;  // debug_fused_trace: stamp [7] first entry, [11] last end of this launch
};

__global__ __launch_bounds__(kDThreads, 2) void conv32ps_kernel(ConvDArgs a,
                                                                ConvTailMap mp,
                                                                ConvStackTab tb) {
  warm_kernargs<sizeof(ConvDArgs) + sizeof(ConvTailMap) + sizeof(ConvStackTab)>();
  const int xcd = blockIdx.x & 7;
  const int r0 = blockIdx.x >> 3;
  const bool main_wg = r0 < mp.mains_per_xcd;
  const int r = main_wg ? r0 : r0 - mp.mains_per_xcd;
  const int c = xcd * (main_wg ? mp.mains_per_xcd : mp.tails_per_xcd) + r;
  if (c >= (main_wg ? mp.n_main : mp.n_tail)) return;
  const long long t0 = a.dbg_wgs ? wall_clock64() : 0;
  const long long t_entry = (tb.stamps && tb.trace) ? wall_clock64() : 0;
  if (tb.stamps && blockIdx.x == 0 && threadIdx.x == 0 && tb.stamps[0]) {
    // publish of the last step -> first instruction of this stack: the host's turn-around
    // + the launch, as the GPU saw it (engine options stat_turn_gpu_ns / stat_turn_count)
    // (steps INSIDE a segment: a turn between two segments -- commit, seed policy, the
    // next init_seed -- takes milliseconds and is not what this figure is about)
    const long long dt = wall_clock64() - tb.stamps[0];
    if (dt < 10000) {
      atomicAdd(reinterpret_cast<unsigned long long*>(tb.stamps + 1), (unsigned long long)dt);
      atomicAdd(reinterpret_cast<unsigned long long*>(tb.stamps + 2), 1ull);
    }
    tb.stamps[0] = 0;
  }
  // (a scalar load in flight under the first conv: no wait of its own)
  const int ahead_ch = tb.ahead_choice ? *tb.ahead_choice : 0;
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
  // what changes from conv to conv, carried in registers from one to the next
  // (rebuilt from the kernel arguments in every loop header it cost two scalar
  // loads and their waits per conv; same bits either way)
  const char* wp = tb.wpack0 + (long)tb.l_begin * tb.wpack_stride;
  const float* bp = tb.bias0 + (long)tb.l_begin * tb.bias_stride;
  const long wstride = tb.wpack_stride, bstride = tb.bias_stride;
  unsigned epoch = tb.epoch0 + (unsigned)tb.l_begin;
  const char* sp_in = (tb.l_begin & 1) ? tb.sp_s : tb.sp_t;
  const char* sp_out = (tb.l_begin & 1) ? tb.sp_t : tb.sp_s;
  const int l_first = tb.l_begin, l_dbg = tb.dbg_layer;
  // Pacing (tb.pace > 0; engine option flow_pace): conv l of a workgroup does not start
  // before  t0 + l pace + phi,  phi = pace_spread x (its first voxel / V), + pace_tail for a
  // tail workgroup: the free-running hand-off lets the lower planes run ahead and every
  // consumer wait for the latest of its ~20 producers; a common beat a little above the
  // chain's own length takes that jitter out (measured: -2.5 % per stack at 7.0 us per
  // conv, profiles/r06_pacing.txt).  Timing only: the arithmetic does not know about it.
  const long long t_pace0 =
      wall_clock64() + (long long)tb.pace_spread * v0 / a.V + (main_wg ? 0 : tb.pace_tail);
  for (int l = tb.l_begin; l < tb.l_end; ++l) {
    if (tb.pace > 0) {
      const long long target = t_pace0 + (long long)(l - l_first) * tb.pace;
      while (wall_clock64() < target) __builtin_amdgcn_s_sleep(1);
    }
    ConvLayer L;
    L.in_sp = sp_in;    // T' for even convs, X' for odd ones ...
    L.out_sp = const_cast<char*>(sp_out);  // ... and the other one written
    L.wpack = wp;
    L.bias = bp;
    L.dbg = l == l_dbg ? a.L.dbg : nullptr;
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
    if (L.dbg) Ldbg = L;
    if (l == l_first + 1 && ahead_ch < 0) {  // (every workgroup alike)
      // ([3]: launches that ended here -- engine statistic stat_ahead_aborted)
      if (tb.stamps && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(tb.stamps + 3), 1ull);
      break;
    }
    const bool last = l == tb.nlayers - 1;
    if (main_wg) {
      const bool dbg_here = blockIdx.x == 0 && a.dbg_wgs != 2;
      if (l == 0 && !(kAbl & 512))
        conv32m_body<1, false, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else if (last && !(kAbl & 512))
        conv32m_body<1, true, true, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else if ((l & 1) && !(kAbl & 256))  // (256: ONE code body for every middle conv;
        // 512: every conv of the stack -- timing only: how much of a layer is
        // instruction delivery when odd and even convs run different code?)
        conv32m_body<0, false, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
      else
        conv32m_body<1, true, false, true, kPsRes>(a, L, 0, v0, gc, dbg_here, xres);
    } else {
      const bool dbg_here = c == 0 && a.dbg_wgs == 2;
      if (l == 0 && !(kAbl & 512))
        conv32d_body<1, false, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                    mp.taoff, dbg_here);
      else if (last && !(kAbl & 512))
        conv32d_body<1, true, kTPieces, true, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                  mp.taoff, dbg_here);
      else if ((l & 1) && !(kAbl & 256))
        conv32d_body<0, false, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                    mp.taoff, dbg_here);
      else
        conv32d_body<1, true, kTPieces, false, 1, kTRows, 2, true>(a, L, 0, v0, gc,
                                                                   mp.taoff, dbg_here);
    }
  }
  stamp_workgroup(a, Ldbg, t0);
  // (debug_fused_trace: [7] first entry, [11] last end of this launch's workgroups)
  if (tb.stamps && tb.trace == 1 && threadIdx.x == 0) {
    atomicMin(reinterpret_cast<unsigned long long*>(tb.stamps + 7), (unsigned long long)t_entry);
    atomicMax(reinterpret_cast<unsigned long long*>(tb.stamps + 11),
              (unsigned long long)wall_clock64());
  }
  // ([25] last end, [26] first entry of the stack IN FRONT of a traced step under stack_ahead)
  // (one plain store per workgroup into its own slot, reduced by the host: 356 atomics on one
  // word at the END of a launch would stand between it and the launch behind it)
  if (tb.stamps && tb.trace == 2 && threadIdx.x == 0) {
    tb.stamps[32 + blockIdx.x] = wall_clock64();
    if (blockIdx.x == 0) tb.stamps[26] = t_entry;
    // ... and where the workgroup ran: (main?, XCC, SE, SH, CU) -- two main workgroups on
    // one CU is what a dispatch onto a chip that is not yet empty can produce
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tb.stamps[32 + 512 + blockIdx.x] =
        (long long)(((main_wg ? 1u : 0u) << 31) | ((xcc & 15u) << 16) | ((hw >> 8) & 0xffu)) + 1;
  }
}

}  // namespace ffn
