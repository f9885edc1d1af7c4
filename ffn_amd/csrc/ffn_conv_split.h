// The split-product 3x3x3 32->32 convs (conv_variant 6 .. 9: conv32d, conv32m, conv32mt), their FLOW hand-off and the helpers they share.
// (part of ffn_kernels.h: included from there, in this order, inside no namespace)
#pragma once

namespace ffn {

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
// it; 64 no dz = +1 DMA; 128 main bodies wait for nobody; 256 / 512 one code body for
// the middle / for all convs of the stack (ffn_conv_resident.h).  (Round 5's table:
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
  if (a.flow_n_main < 0)  // conv32h: uniform 80-voxel producers (d / 80)
    return (int)__umulhi((unsigned)d, 53687092u);
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
      // word 1 names the CAUSE for this very step (the faces launch / ffn_predict
      // read it next to word 0): a time-out, not the fp16 range check
      __hip_atomic_store(vflag + 1, a.range_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

}  // namespace ffn
