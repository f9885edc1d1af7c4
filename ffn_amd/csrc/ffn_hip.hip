// libffn_hip.so -- host side of the C-ABI declared in include/ffn_hip.h.
//
// One engine = one GPU + one HIP stream + the conv-stack weights + staging and
// activation buffers for up to max_batch concurrent fields of view.  Canvases
// (image / seed / segmentation of a whole subvolume) are device resident.
// The reference interfaces each entry point replaces are cited in the header.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ffn_internal.h"
#include "ffn_kernels.h"
#include "ffn_host_loop.h"

using namespace ffn;

namespace {

constexpr int kHeadBlocks = 281;  // head kernel grid.x (4 sweeps of 32 voxels x 281)

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

}  // namespace

int ffn_set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

namespace {

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess)                                                    \
      return fail(FFN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                  hipGetErrorString(_e), __FILE__, __LINE__);                \
  } while (0)

}  // namespace

struct ffn_engine {
  // Entry points may be called from several host threads (the two canvas groups
  // of MultiCanvasDriver, their between-segment work): each one holds this lock
  // while it touches engine state or queues work on the stream -- except the
  // wait for a step's completion flags, which is what the threads overlap.
  std::recursive_mutex mu;
  // The canvas utility calls (point / box reads and writes, commit counts) own
  // the scratch buffers and WAIT for their result; they hold `util_mu` for all
  // of it but `mu` only while they queue their work, and wait for their own
  // event, not for the stream -- so another thread's next FoV step is queued
  // (and runs) behind them instead of waiting for the host to wake up.
  std::mutex util_mu;
  // They run on their OWN stream: a canvas between two segments is in no step,
  // so what is committed / read / re-seeded on it must not queue behind the FoV
  // steps other canvases have in flight (2 ms each at batch 16).  Ordering
  // per canvas: the step side waits for the canvas' last utility event
  // (ffn_canvas::ev_util) before its next step; the utility side starts after
  // the canvas' last step has pasted -- batched steps paste BEFORE the faces
  // kernel raises the completion flag the host waits for, a single-FoV step
  // (faces first: the host's turn-around is on its critical path) makes the
  // utility stream wait for an event recorded behind it.
  hipStream_t ustream = nullptr;
  hipEvent_t main_ev = nullptr;  // "everything queued on `stream` so far"
  // ffn_canvas_segment_many_carry: the batched step it left in flight when it
  // returned (one per engine: the driver that runs ONE group of canvases and
  // does a canvas' between-segment work under the others' next step).  `many_mu`
  // is taken before `mu`, never the other way round.
  std::mutex many_mu;
  ffn_host::ManyCarry many_carry;
  uint32_t many_ticket = 0;
  std::vector<ffn_canvas*> many_canvases;  // the carried step's, in batch order
  int device = 0;
  hipStream_t stream = nullptr;
  Geom g{};   // the FoV as the caller sees it (zyx): gather / paste / faces, I/O
  Geom gp{};  // the FoV as the split-product kernels (conv_variant >= 6) lay it
              // out: the same, or with its axes permuted (oa) so that the
              // SHORTEST axis is the row direction -- rows of 21 instead of 41
              // voxels put the anisotropic 21 x 41 x 41 FoV inside the row
              // budget of conv32m / conv32mt
  bool permuted = false;
  int depth = 0;
  int max_batch = 0;
  bool weights_set = false;

  float* act_base = nullptr;  // 2 activation buffers (T, X) x max_batch
  float* bufT = nullptr;      // logical origins
  float* bufX = nullptr;
  uint32_t* validbits = nullptr;
  int32_t* pidx = nullptr;    // dense FoV index -> padded position (variant 2)
  int nchunks_c = 0, Rc = 0;
  int fuse_head = 1;      // 1x1x1 head fused into the last conv32c launch
  int count_blocks = kHeadBlocks;  // entries per item in `count` for the last step
  int store_policy = 1;  // conv32c epilogue stores: sc1 write-through (-1.4 % per stack)
  long long* d_dbg = nullptr;  // debug clocks of conv32c WG 0 (24 values)
  int dbg_clock = 0;
  int dbg_layer = 3;      // the launch whose clocks are recorded (3 = a conv_a)
  size_t lds_bytes_c = 0;
  // conv32d (variant 6): 160-voxel chunks, the 27 taps split over the waves,
  // split-plane activations, LDS-DMA staging
  int nchunks_k = 0, Rc_k = 0;
  bool k_ok = false;             // the chunk + halo fits the LDS slots
  // rawT / rawX / rawS: the three activation allocations
  float* rawT = nullptr;
  float* rawX = nullptr;
  float* rawS = nullptr;
  bool d_ok = false;
  // conv32d with 96-voxel chunks, two workgroups per CU (variant 7)
  bool e_ok = false;
  int batch_chunks = 1;   // option: variant 6 uses the 96-voxel form for n >= 2
  bool small_now = false; // the stack being queued uses the 96-voxel form
  // conv32m (variant 8): M split over the waves, weights through an LDS ring
  bool m_ok = false;
  bool m_now = false;
  // conv32mt (variant 9): conv32m for the first n_main <= 256 chunks, 32-voxel
  // K-split tail workgroups for the rest
  bool t_ok = false;
  bool t_now = false;
  int tail_batched = 0;  // option: steps with >= 2 FoVs take the tail form too
  // FLOW (ffn_conv_split.h "flagged launches"): 0 off; 1 conv32mt's launches with
  // the flagged hand-off compiled in (still one dependent launch per conv: the
  // words are always there already -- isolates the cost of the sc1 reads and
  // the poll); 2 the resident stack, conv32ps: ONE launch for the 2 depth - 1 convs
  // of a single-FoV step
  int flow = 0;
  unsigned* flow_flags = nullptr;  // one word per producer workgroup, kFlowStride apart
  unsigned* flow_err = nullptr;    // polls that gave up, ever
  unsigned flow_epoch = 0;         // sequence number of the last conv queued
  int flow_debug = 0;              // debug option: ConvDArgs::flow_dbg
  // The resident launch needs every one of its workgroups on the chip at once.
  // flow_fits: the device can hold them (CU count x occupancy, checked at create);
  // at run time a poll that gives up voids the step (FFN_ERR_FLOW): the repeat
  // runs per-layer launches (flow_skip, one stack), and kFlowStrikes voided
  // resident steps in a row turn the resident launch off for this engine
  // (flow_auto_off; option "flow" turns it on again).
  bool flow_fits = false;
  int flow_strikes = 0;            // voided resident steps since the last good one
  int flow_skip = 0;               // the next single-FoV stack runs per-layer launches
  int flow_auto_off = 0;
  long stat_flow_voids = 0;
  bool last_stack_resident = false;  // what run_stack queued last
  bool slot_resident[2] = {false, false};
  long long* flow_trace = nullptr; // debug_clock 4: ConvDArgs::flow_trace
  int flow_trace_slots = 0;
  int n_main = 0, n_tail = 0, n_tail3 = 0;  // tail workgroups of 32 / 96 voxels
  int tsched_aoff[4 * 8] = {};
  int t3sched_aoff[4 * 8] = {};
  int nchunks_m = 0;
  int nchunks_e = 0;
  size_t lds_bytes_e = 0;
  int esched_aoff[4 * 8] = {};
  bool d_weights_ok = true;      // every |weight| x 2^11 inside the fp16 range
  uint16_t* wpackd = nullptr;    // [28][khalf][plane hi, res][64][8] fp16 per layer
  size_t wpackd_layer = 0;       // halves per layer
  // conv32h / conv32hs (variant 10): 80-voxel workgroups on 16x16x32 tiles, two
  // hand-off chains per SIMD for a single FoV; several FoVs run conv32m as under 9
  uint16_t* wpackh = nullptr;    // [28][out half][plane hi, res][64][8] fp16 per layer
  bool h_ok = false;             // geometry + residency
  bool h_now = false;            // the stack being queued is conv32h's
  int n_half = 0;                // 80-voxel chunks of the FoV
  // Pacing of the resident stack (ConvStackTab::pace, 10-ns ticks between two convs of a
  // workgroup).  flow_pace: -1 = the beat measured by tune_pace() when the weights were
  // set (conv_variant 9; 0 if no beat beat the free-running stack), 0 = off, > 0 = that
  // beat.  flow_pace_spread: -1 = as wide as the beat (the FoV's last voxel one beat behind
  // its first), else ticks.
  int flow_pace = -1;
  int flow_pace_tail = 0;        // ConvStackTab::pace_tail
  int flow_pace_spread = -1;
  int pace_auto = 0;             // tune_pace()'s beat (0: none)
  float pace_auto_us[2] = {0.f, 0.f};  // us per stack it measured: free-running, at the beat
  int cus = 0;                   // compute units of the device
  size_t lds_bytes_d = 0;        // 3 slots x 8 planes x Rc_k rows x 16 B
  int dsched_aoff[4 * 8] = {};
  int dsched_btap[4 * 8] = {};
  unsigned* range_flag = nullptr;  // device word: tag of the last void run
  // Speculative conv0_a of a single-FoV step's successor (SpecArgs): the launch
  // queued behind the last step's paste, and what a step must match to run on it
  struct Spec {
    bool valid = false;
    const ffn_canvas* canvas = nullptr;
    int n = 0;
    int pos[kSpecMax][3] = {};
    float pad_value = 0.f, move_thr = 0.f;
    int variant = 0;
  } spec;
  int speculate = 1;        // option
  int spec_force_mismatch = 0;  // debug option: fail the next N matches
  long stat_spec_mismatch = 0;
  long stat_many_carried = 0;  // batched steps left in flight across a return
  int fuse_paste = 1;       // option: faces + paste of a single FoV as one launch
  int fuse_conv0a = 1;      // option: ... and the next step's conv0_a in it as well
  int* d_spec_choice = nullptr;
  long stat_spec_launched = 0, stat_spec_hits = 0;
  long stat_spec_miss_full = 0, stat_spec_miss_short = 0;
  // ffn_canvas_segment_turn on the host: queueing its sequence, waiting for its record
  long long stat_segturn_queue_ns = 0, stat_segturn_wait_ns = 0;
  long stat_segturn_calls = 0;
  // The NEXT step's resident stack queued right behind the launch that holds its
  // speculative conv0_a, before the host has seen this step's record (engine option
  // stack_ahead): the stack reads only what that conv0_a wrote (and gives up after its
  // first conv when the device found no valid position), so the host's turn-around and
  // the launch latency of the stack leave the step's critical path.
  int stack_ahead = 1;
  int paste_blocks = 0;  // fused step launch: paste blocks (0: one block per CU, see kPasteBlocks)
  int debug_submit_delay_ns = 0;
  int debug_fused_twice = 0;  // experiment: the host idles this long in front of a step's launches
  bool trace_now = false;     // debug_fused_trace: the launches being queued stamp
  bool ahead_valid = false;   // such a stack is in the stream, for the step e->spec describes
  long stat_ahead_used = 0, stat_ahead_wasted = 0;
  unsigned range_tag = 0;        // tag of the run being queued
  bool fp16_ok = true;           // every weight inside the fp16 range
  int conv_variant = 0;       // 0 conv32 (any FoV), 2 conv32c (exact f32), 6 conv32d, 7 = 6
                              // with 96-voxel chunks, 8 conv32m, 9 conv32mt (+ conv32m)
  int exact_variant = 0;      // the f32 kernel a voided fp16 step is repeated with: 2
                              // where conv32c takes the FoV, else 0
  float* h_io = nullptr;      // pinned staging of ffn_predict: seed, image, logits
  float* up_image = nullptr;  // dense FoVs uploaded by ffn_predict
  float* up_seed = nullptr;
  float* seed_raw = nullptr;  // raw (NaN-preserving) seed FoV of the current step
  // The conv0_a of the NEXT step may run in the same launch as this step's faces
  // and paste (faces_paste_conv0a_kernel): what it writes -- the raw seed copy,
  // a range flag, the position it chose -- goes to the OTHER of two sets, which
  // run_stack makes the current one when that step is queued.
  float* seed_raw_alt = nullptr;
  unsigned* range_flag_alt = nullptr;
  int* d_spec_choice_alt = nullptr;
  float* logits = nullptr;
  unsigned* count = nullptr;
  uint8_t* valid = nullptr;
  float* weights = nullptr;  // one allocation; layout below
  size_t w0a_off = 0, b0a_off = 0, wl_off = 0;
  size_t w0ap_off = 0;  // conv0_a weights with the taps in gp's axis order
  std::vector<size_t> wpack_off, bias_off;  // 2*depth-1 entries (conv0_b ..)

  StepItem* d_items = nullptr;
  StepItem* h_items = nullptr;
  // pinned, written by the faces block: kPubWords 8-byte words per item and slot,
  // (step number << 32) | one 32-bit word of the item's ffn_step_result
  unsigned long long* h_pub = nullptr;
  unsigned step_id = 0;
  // two result / descriptor slots: one step may be queued behind the running one
  int next_slot = 0;
  int slot_n[2] = {0, 0};
  bool slot_waited[2] = {false, false};  // a thread is in ffn_canvas_step_wait for it
  unsigned slot_ticket[2] = {0, 0};
  std::vector<ffn_canvas*> slot_canvas[2];
  int sync_mode = 1;  // 0 = hipStreamSynchronize, 1 = poll h_pub (then sync)
  // since the last set_option("stat_reset"): batched step calls, FoVs in them,
  // and a histogram of FoVs per call (index min(n, 64))
  long stat_calls = 0, stat_items = 0;
  long stat_hist[65] = {};
  // the turn-around between two single-FoV steps: the GPU's view (d_stamps, see
  // ConvStackTab::stamps) and the host's (steady-clock ns from "record arrived" to
  // "the next resident launch is queued", and inside that hipLaunchKernelGGL alone)
  long long* d_stamps = nullptr;
  long long t_arrived_ns = 0;
  int fused_trace_in = 0;        // debug_fused_trace: steps until the traced one
  long long stat_turn_host_ns = 0, stat_launch_host_ns = 0;
  long stat_turn_host_count = 0;

  void* d_scratch = nullptr;
  void* h_scratch = nullptr;
  size_t scratch_bytes = 0;

  int prof_mode = 0;
  std::vector<hipEvent_t> events;
  int events_used = 0;
  // the event pair around a stack queued AHEAD (stack_ahead): it joins the samples when the
  // stack is used by its step -- one that found no position to run at ends after its first
  // conv, and its ~10 us would pass for a stack's duration
  hipEvent_t ahead_ev[2] = {nullptr, nullptr};
  bool ahead_ev_pending = false;
  double conv_ms = 0.0;
  int64_t conv_launches = 0;
  size_t lds_bytes = 0;
  std::vector<ffn_canvas*> canvases;  // live canvases created from this engine
  int prof_every = 1;                 // profile 1 of every N run_stack calls
  long stack_calls = 0;
  bool prof_now = false;
  int ablate = 0;                     // debug: skip phases of the conv kernel
  std::vector<int> chain_launches_pending;  // mode 2: launches per event pair
  std::vector<float> prof_samples;  // ms of every event pair since the last reset
};

struct ffn_canvas {
  ffn_engine* engine = nullptr;
  float* image = nullptr;        // f32 image; NULL for a uint8 canvas
  uint8_t* image_u8 = nullptr;   // uint8 canvas: raw image (1 B / voxel) ...
  float* image_lut = nullptr;    // ... and (v - mean) / stddev for v = 0..255
  float* seed = nullptr;
  int32_t* seg = nullptr;
  int cz = 0, cy = 0, cx = 0;
  size_t nvox = 0;
  // Bounding box of every seed voxel that may differ from NaN since the last
  // clear: Canvas.init_seed (inference.py:443-450) clears the WHOLE volume per
  // seed (62.5 MB at 250^3, 4.3 GB at 1024^3); here only this box is re-filled.
  int dirty_lo[3] = {0, 0, 0};
  int dirty_hi[3] = {0, 0, 0};  // exclusive; lo >= hi: nothing dirty
  ffn_host::SegmentState loop;  // ffn_canvas_segment_at's queue / visited set
  hipEvent_t ev_util = nullptr;  // behind the last utility operation on this canvas
  bool util_pending = false;     // ... which the next step has to wait for
  bool paste_after_flag = false; // the last step pasted AFTER raising its flag
  // the segment loop's guess of the positions the step after the next one may
  // be made at (consumed by the next single-FoV ffn_canvas_step_submit)
  int hint_n = 0;
  int hint_pos[kSpecMax][3] = {};
  bool hint_from_loop = false;  // the next step is the segment loop's own

  void mark_dirty(const int lo[3], const int hi[3]) {
    const int dims[3] = {cz, cy, cx};
    const bool empty = dirty_lo[0] >= dirty_hi[0];
    for (int k = 0; k < 3; ++k) {
      const int l = std::max(lo[k], 0), h = std::min(hi[k], dims[k]);
      dirty_lo[k] = empty ? l : std::min(dirty_lo[k], l);
      dirty_hi[k] = empty ? h : std::max(dirty_hi[k], h);
    }
  }
};

namespace {
inline long long steady_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline unsigned next_tag(unsigned t) { return t + 1 ? t + 1 : 1; }  // never 0
// A speculative conv0_a launch that no step will use: its range tag is spent.
inline void drop_spec(ffn_engine* e) {
  if (e->spec.valid && !e->ahead_valid) e->range_tag = next_tag(e->range_tag);
  // (a stack queued ahead has taken the tag already: run_stack)
  if (e->ahead_valid) e->stat_ahead_wasted += 1;
  e->spec.valid = false;
  e->ahead_valid = false;
}
struct EngineLock {
  std::unique_lock<std::recursive_mutex> lk;
  explicit EngineLock(ffn_engine* e) {
    if (e) lk = std::unique_lock<std::recursive_mutex>(e->mu);
  }
};
constexpr int kFlowStrikes = 3;

// A step came back void, and its own record says why (ffn_step_result.range_error
// 3, written from word 1 of the step's range flag): the resident launch gave up on
// a producer (conv32ps: a poll reached its bound -- some workgroup was not on the
// chip, or far too late), not the fp16 range check.  Then the caller gets
// FFN_ERR_FLOW instead of FFN_ERR_RANGE: same contract (nothing was pasted,
// repeat the step), but the arithmetic stays -- the repeat runs the same convs as
// per-layer launches (same bits) -- and after kFlowStrikes such steps in a row
// the engine stops using the resident launch.  0: not a flow time-out.
int flow_voided(ffn_engine* e, bool timed_out) {
  if (!timed_out) return 0;
  EngineLock lock_(e);
  e->stat_flow_voids += 1;
  e->flow_strikes += 1;
  e->flow_skip = 1;
  if (e->flow_strikes >= kFlowStrikes && e->flow == 2) {
    e->flow = 0;
    e->flow_auto_off = 1;
    e->flow_skip = 0;
    return fail(FFN_ERR_FLOW,
                "the resident conv launch timed out waiting for one of its own "
                "workgroups %d steps in a row (is the GPU shared or partitioned?): "
                "the step changed nothing; this engine now runs one launch per conv "
                "(option flow = 0; set flow = 2 to try again) -- repeat the step",
                kFlowStrikes);
  }
  return fail(FFN_ERR_FLOW,
              "the resident conv launch timed out waiting for one of its own "
              "workgroups: the step changed nothing; repeat it (the repeat runs one "
              "launch per conv, same arithmetic)");
}

// util_mu, then mu (the step path takes only mu: no lock-order inversion)
struct UtilLock {
  std::unique_lock<std::mutex> ul;
  std::unique_lock<std::recursive_mutex> lk;
  explicit UtilLock(ffn_engine* e) {
    if (e) {
      ul = std::unique_lock<std::mutex>(e->util_mu);
      lk = std::unique_lock<std::recursive_mutex>(e->mu);
    }
  }
  // Before the first utility operation of a call: the canvas' last step has
  // pasted (see ffn_engine::ustream).
  // reads_only: the call looks at the canvas and changes nothing -- a speculative conv0_a
  // (and the stack queued behind it) stays what it is; the segment loop's own point reads
  // between two steps used to cost it its launch (0.5 % of the steps: a whole stack run for
  // nothing + a step made the ordinary way)
  hipError_t begin(ffn_engine* e, ffn_canvas* c, bool reads_only = false) {
    if (!reads_only) drop_spec(e);  // the canvas may change under a speculative conv0_a
    if (!c->paste_after_flag) return hipSuccess;
    hipError_t err = hipEventRecord(e->main_ev, e->stream);
    if (err == hipSuccess) err = hipStreamWaitEvent(e->ustream, e->main_ev, 0);
    if (err == hipSuccess) c->paste_after_flag = false;
    return err;
  }
  // After the last one: mark the point the canvas' next step waits for ...
  hipError_t end(ffn_engine* e, ffn_canvas* c) {
    c->util_pending = true;
    return hipEventRecord(c->ev_util, e->ustream);
  }
  // ... and, for calls that return data, wait for it (`mu` released meanwhile)
  hipError_t wait(ffn_engine* e, ffn_canvas* c) {
    hipError_t err = end(e, c);
    if (err != hipSuccess) return err;
    lk.unlock();
    err = hipEventSynchronize(c->ev_util);
    lk.lock();
    if (err == hipSuccess) c->util_pending = false;  // done: nothing to wait for
    return err;
  }
};
// host-side: nothing queued on canvas c is still running
hipError_t canvas_quiesce(ffn_engine* e, ffn_canvas* c) {
  drop_spec(e);  // the caller is about to read or change the canvas directly
  hipError_t err = hipSuccess;
  if (c->paste_after_flag) {
    err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) c->paste_after_flag = false;
  }
  if (err == hipSuccess && c->util_pending) {
    err = hipEventSynchronize(c->ev_util);
    if (err == hipSuccess) c->util_pending = false;
  }
  return err;
}
}  // namespace

int ffn_canvas_view(ffn_canvas* c, FfnCanvasView* out) {
  if (!c || !out) return ffn_set_error(FFN_ERR_ARG, "NULL canvas");
  if (!c->engine)
    return ffn_set_error(FFN_ERR_STATE, "canvas outlived its engine");
  {
    // the caller (label / seed kernels on their own streams) reads the canvas
    // once the engine's stream is idle: the utility stream has to be, too
    EngineLock lock_(c->engine);
    if (canvas_quiesce(c->engine, c) != hipSuccess)
      return ffn_set_error(FFN_ERR_HIP, "canvas_quiesce failed");
  }
  out->device_id = c->engine->device;
  out->engine_stream = c->engine->stream;
  out->image = c->image;
  out->image_u8 = c->image_u8;
  out->image_lut = c->image_lut;
  out->segmentation = c->seg;
  out->shape_zyx[0] = c->cz;
  out->shape_zyx[1] = c->cy;
  out->shape_zyx[2] = c->cx;
  return FFN_OK;
}

namespace {

int ensure_scratch(ffn_engine* e, size_t bytes) {
  if (bytes <= e->scratch_bytes) return FFN_OK;
  HIP_TRY(hipStreamSynchronize(e->ustream));  // (only utility calls use it)
  if (e->d_scratch) HIP_TRY(hipFree(e->d_scratch));
  if (e->h_scratch) HIP_TRY(hipHostFree(e->h_scratch));
  e->d_scratch = e->h_scratch = nullptr;
  e->scratch_bytes = 0;
  size_t want = std::max<size_t>(bytes, 1 << 20);
  HIP_TRY(hipMalloc(&e->d_scratch, want));
  HIP_TRY(hipHostMalloc(&e->h_scratch, want, hipHostMallocDefault));
  e->scratch_bytes = want;
  return FFN_OK;
}

template <bool RI, bool RO, bool SK>
int set_lds_attr(size_t bytes) {
  HIP_TRY(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&conv32_kernel<RI, RO, SK>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return FFN_OK;
}

// x ~= hi + 2^-11 * res with hi, res fp16 (what the device does when staging)
inline void split_fp16x2(float x, uint16_t part[2]) {
  const float xh = std::fabs(x) < 6.103515625e-05f ? 0.0f : x;
  const _Float16 hi = (_Float16)xh;
  const _Float16 res = (_Float16)((x - (float)hi) * 2048.0f);
  std::memcpy(&part[0], &hi, 2);
  std::memcpy(&part[1], &res, 2);
}

int set_lds_attr_d(size_t bytes) {
#define FFN_D_ATTR(KIND, SK, KSV, HEADV)                                          \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32d_kernel<KIND, SK, KSV, HEADV>),       \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes))
#define FFN_D_ATTRS(KSV)            \
  FFN_D_ATTR(0, false, KSV, false); \
  FFN_D_ATTR(1, false, KSV, false); \
  FFN_D_ATTR(1, true, KSV, false);  \
  FFN_D_ATTR(1, false, KSV, true);  \
  FFN_D_ATTR(1, true, KSV, true)
  FFN_D_ATTRS(8);
  FFN_D_ATTRS(9);
  FFN_D_ATTRS(10);
#undef FFN_D_ATTRS
#undef FFN_D_ATTR
  return FFN_OK;
}

constexpr int kERows = 208, kETiles = 3, kEPieces = 7;

int set_lds_attr_m() {
#define FFN_M_ATTR(KIND, SK, HEADV)                                               \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32m_kernel<KIND, SK, HEADV>),            \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMLdsBytes))
  FFN_M_ATTR(0, false, false);
  FFN_M_ATTR(1, false, false);
  FFN_M_ATTR(1, true, false);
  FFN_M_ATTR(1, false, true);
  FFN_M_ATTR(1, true, true);
#undef FFN_M_ATTR
#define FFN_MT_ATTR(KIND, SK, HEADV)                                              \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32mt_kernel<KIND, SK, HEADV, 1>),        \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMLdsBytes));              \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32mt_kernel<KIND, SK, HEADV, 3>),        \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMLdsBytes))
  FFN_MT_ATTR(0, false, false);
  FFN_MT_ATTR(1, false, false);
  FFN_MT_ATTR(1, true, false);
  FFN_MT_ATTR(1, false, true);
  FFN_MT_ATTR(1, true, true);
#undef FFN_MT_ATTR
#define FFN_MTF_ATTR(KIND, SK, HEADV)                                             \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32mt_kernel<KIND, SK, HEADV, 1, true>),  \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMLdsBytes))
  FFN_MTF_ATTR(0, false, false);
  FFN_MTF_ATTR(1, false, false);
  FFN_MTF_ATTR(1, true, false);
  FFN_MTF_ATTR(1, true, true);
#undef FFN_MTF_ATTR
#define FFN_H_ATTR(KIND, SK, HEADV)                                               \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(&conv32h_kernel<KIND, SK, HEADV>),            \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHLdsBytes))
  FFN_H_ATTR(0, false, false);
  FFN_H_ATTR(1, false, false);
  FFN_H_ATTR(1, true, false);
  FFN_H_ATTR(1, true, true);
#undef FFN_H_ATTR
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32hs_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kHLdsBytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32ps_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kMLdsBytes));
  return FFN_OK;
}

int set_lds_attr_e(size_t bytes) {
#define FFN_E_ATTR(KIND, SK, HEADV)                                               \
  HIP_TRY(hipFuncSetAttribute(                                                    \
      reinterpret_cast<const void*>(                                              \
          &conv32d_kernel<KIND, SK, kEPieces, HEADV, kETiles, kERows, 2>),        \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes))
  FFN_E_ATTR(0, false, false);
  FFN_E_ATTR(1, false, false);
  FFN_E_ATTR(1, true, false);
  FFN_E_ATTR(1, false, true);
  FFN_E_ATTR(1, true, true);
#undef FFN_E_ATTR
  return FFN_OK;
}

template <bool RI, bool RO, bool SK, int DBG = 0>
int set_lds_attr_c(size_t bytes) {
  HIP_TRY(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&conv32c_kernel<RI, RO, SK, DBG, 8>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (DBG == 0)
    HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv32c_kernel<RI, RO, SK, 0, 9>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (DBG == 0 && SK) {  // the fused-head form of the last conv
    HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv32c_kernel<RI, RO, SK, 0, 8, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv32c_kernel<RI, RO, SK, 0, 9, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return FFN_OK;
}

// Flush the per-launch event pairs recorded since the last flush.
int flush_events(ffn_engine* e) {
  if (e->events_used == 0) return FFN_OK;
  HIP_TRY(hipEventSynchronize(e->events[e->events_used - 1]));
  for (int k = 0; k + 1 < e->events_used; k += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->events[k], e->events[k + 1]));
    e->conv_ms += ms;
    if (e->prof_samples.size() < (size_t)(1 << 16)) e->prof_samples.push_back(ms);
    if (e->prof_mode == 2 && !e->chain_launches_pending.empty()) {
      e->conv_launches += e->chain_launches_pending.front();
      e->chain_launches_pending.erase(e->chain_launches_pending.begin());
    } else {
      e->conv_launches += 1;
    }
  }
  e->events_used = 0;
  return FFN_OK;
}

template <bool RI, bool RO, bool SK>
int launch_conv32(ffn_engine* e, int n, const float* in, float* out,
                  const float* skip, int layer) {
  ConvArgs a;
  a.in = in;
  a.out = out;
  a.skip = skip;
  a.wpack = e->weights + e->wpack_off[layer];
  a.bias = e->weights + e->bias_off[layer];
  a.valid = e->valid;
  a.act_stride = e->g.act_stride;
  a.XS = e->g.XS;
  a.plane = e->g.plane;
  a.R = e->g.R;
  a.nchunks = e->g.nchunks;
  const bool prof = e->prof_now;
  if (prof) {
    if (e->events_used + 2 > (int)e->events.size()) {
      int rc = flush_events(e);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  }
  hipLaunchKernelGGL((conv32_kernel<RI, RO, SK>), dim3(n * e->g.nchunks),
                     dim3(kConvThreads), e->lds_bytes, e->stream, a);
  if (prof) HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  return FFN_OK;
}

struct HeadFusion {
  bool on = false;
  float pad_value = 0.f, move_thr = 0.f;
};

template <bool RI, bool RO, bool SK>
int launch_conv32c(ffn_engine* e, int n, const float* in, float* out,
                   const float* skip, int layer,
                   const HeadFusion& head = HeadFusion()) {
  ConvCArgs a;
  a.in = in;
  a.out = out;
  a.skip = skip;
  a.wpack = e->weights + e->wpack_off[layer];
  a.bias = e->weights + e->bias_off[layer];
  a.pidx = e->pidx;
  a.act_stride = e->g.act_stride;
  a.XS = e->g.XS;
  a.plane = e->g.plane;
  a.Rc = e->Rc;
  a.nchunks = e->nchunks_c;
  a.V = e->g.V;
  a.fx = e->g.fx;
  a.fyfx = e->g.fy * e->g.fx;
  a.total_slots = n * e->nchunks_c;
  a.slots_per_xcd = (a.total_slots + 7) / 8;
  a.nbytes = (unsigned)((size_t)e->g.nchunks * kChunk * kFeatures * sizeof(float));
  a.store_policy = e->store_policy;
  // clocks of ONE mid-stack launch (a conv_a: ReLU in and out, no residual)
  a.dbg = (e->dbg_clock && layer == e->dbg_layer) ? e->d_dbg : nullptr;
  a.head_w = e->weights + e->wl_off;
  a.seed_raw = e->seed_raw;
  a.logits = e->logits;
  a.head_count = e->count;
  a.pad_value = head.pad_value;
  a.move_thr = head.move_thr;
  const bool prof = e->prof_now;
  if (prof) {
    if (e->events_used + 2 > (int)e->events.size()) {
      int rc = flush_events(e);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  }
  const dim3 grid(8 * a.slots_per_xcd), block(kConvThreads);
  a.range_flag = e->range_flag;
  a.range_tag = e->range_tag;
  if (RI == false && RO == false && SK == true && e->ablate != 0 &&
      e->Rc == 256) {
    switch (e->ablate) {  // issue-rate experiments (conv_b instantiation only)
      case 8:
        hipLaunchKernelGGL((conv32c_kernel<false, false, true, 8>), grid, block,
                           e->lds_bytes_c, e->stream, a);
        break;
      case 16:
        hipLaunchKernelGGL((conv32c_kernel<false, false, true, 16>), grid, block,
                           e->lds_bytes_c, e->stream, a);
        break;
      case 24:
        hipLaunchKernelGGL((conv32c_kernel<false, false, true, 24>), grid, block,
                           e->lds_bytes_c, e->stream, a);
        break;
      default:
        return fail(FFN_ERR_ARG, "unsupported ablate mask %d for variant 2",
                    e->ablate);
    }
  } else if (head.on) {
    if (e->Rc == 256)
      hipLaunchKernelGGL((conv32c_kernel<RI, RO, SK, 0, 8, true>), grid, block,
                         e->lds_bytes_c, e->stream, a);
    else
      hipLaunchKernelGGL((conv32c_kernel<RI, RO, SK, 0, 9, true>), grid, block,
                         e->lds_bytes_c, e->stream, a);
  } else if (e->Rc == 256) {
    hipLaunchKernelGGL((conv32c_kernel<RI, RO, SK, 0, 8>), grid, block,
                       e->lds_bytes_c, e->stream, a);
  } else {
    hipLaunchKernelGGL((conv32c_kernel<RI, RO, SK, 0, 9>), grid, block,
                       e->lds_bytes_c, e->stream, a);
  }
  if (prof) HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  return FFN_OK;
}

// The split-plane layout of variant 6 and the f32 layout of the others keep their
// zero padding in different bytes of the same allocations: re-zero on a change.
int switch_variant(ffn_engine* e, int value) {
  if ((value >= 6) != (e->conv_variant >= 6)) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(e->act_base, 0,
                           (size_t)3 * e->max_batch * e->g.act_stride * sizeof(float),
                           e->stream));
  }
  e->conv_variant = value;
  return FFN_OK;
}

// conv32d launch: KIND 0 conv_a (T' = split(relu(conv(X') + b))), KIND 1 conv_b
// (X = conv(T') + b [+ X]; X' = split(relu(X))), or the fused head
void conv32d_args(ffn_engine* e, int n, const float* raw_in, float* raw_out, int layer,
                  const HeadFusion& head, ConvDArgs& a) {
  const Geom& g = e->gp;
  const long positions = g.act_stride / kFeatures;
  a.L.in_sp = reinterpret_cast<const char*>(raw_in) + (size_t)g.guard * 16;
  a.L.out_sp = reinterpret_cast<char*>(raw_out) + (size_t)g.guard * 16;
  a.x_f32 = e->rawX + (size_t)g.guard * 4;
  a.L.wpack = reinterpret_cast<const char*>(e->wpackd + (size_t)layer * e->wpackd_layer);
  a.L.bias = e->weights + e->bias_off[layer];
  a.item_bytes = g.act_stride * (long)sizeof(float);
  a.sp_plane_bytes = positions * 16;
  a.XS = g.XS;
  a.plane = g.plane;
  const bool small = e->small_now;  // 96-voxel chunks, 2 workgroups / CU
  const bool msplit = e->m_now;     // conv32m: 128-voxel chunks, M split
  const int nchunks = msplit ? e->nchunks_m : small ? e->nchunks_e : e->nchunks_k;
  a.nchunks = nchunks;
  a.V = g.V;
  a.fx = g.fx;
  a.fyfx = g.fy * g.fx;
  a.total_slots = n * nchunks;
  a.slots_per_xcd = (a.total_slots + 7) / 8;
  auto magic = [](int d) { return (unsigned)(((1ull << 32) + d - 1) / d); };
  a.magic_nchunks = magic(nchunks);
  a.magic_fyfx = magic(g.fy * g.fx);
  a.magic_fx = magic(g.fx);
  a.permuted = e->permuted;
  a.ds0 = g.dstr[0];
  a.ds1 = g.dstr[1];
  a.ds2 = g.dstr[2];
  a.sp_bytes = (unsigned)((size_t)g.act_stride * sizeof(float) - (size_t)g.guard * 32);
  std::memcpy(a.aoff, small ? e->esched_aoff : e->dsched_aoff, sizeof(a.aoff));
  std::memcpy(a.btap, e->dsched_btap, sizeof(a.btap));
  a.L.dbg = (e->dbg_clock && layer == e->dbg_layer) ? e->d_dbg : nullptr;
  a.dbg_wgs = e->dbg_clock == 2 ? 1 : e->dbg_clock == 3 ? 2 : 0;
  a.head_w = e->weights + e->wl_off;
  a.seed_raw = e->seed_raw;
  a.logits = e->logits;
  a.head_count = e->count;
  a.pad_value = head.pad_value;
  a.move_thr = head.move_thr;
  a.range_flag = e->range_flag;
  a.range_tag = e->range_tag;
  a.flow_flags = e->flow_flags;
  a.flow_n_main = e->n_main;
  a.flow_err = e->flow_err;
  a.flow_halo = g.fy * g.fx + g.fx + 1;
  a.flow_dbg = e->flow_debug;
  a.flow_trace = e->dbg_clock == 4 ? e->flow_trace : nullptr;
  a.L.layer = layer;
  a.L.flow_wait_on = 0;
  a.L.flow_wait = a.L.flow_set = 0;
}

void tail_map(const ffn_engine* e, int n, ConvTailMap& mp) {
  const bool one = n == 1;
  mp.n = n;
  mp.n_main = e->n_main;
  mp.n_tail = one ? e->n_tail : e->n_tail3;
  mp.mains_per_xcd = (mp.n_main + 7) / 8;
  mp.tails_per_xcd = (mp.n_tail + 7) / 8;
  std::memcpy(mp.taoff, one ? e->tsched_aoff : e->t3sched_aoff, sizeof(mp.taoff));
}

template <int KIND, bool SK>
int launch_conv32d(ffn_engine* e, int n, const float* raw_in, float* raw_out,
                   int layer, const HeadFusion& head = HeadFusion()) {
  ConvDArgs a;
  conv32d_args(e, n, raw_in, raw_out, layer, head, a);
  const bool small = e->small_now;
  const bool msplit = e->m_now;
  const bool prof = e->prof_now;
  if (prof) {
    if (e->events_used + 2 > (int)e->events.size()) {
      int rc = flush_events(e);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  }
  const dim3 grid(8 * a.slots_per_xcd), block(kDThreads);
  const int ks = e->Rc_k / 32;
#define FFN_D_LAUNCH(KSV, HEADV)                                                 \
  hipLaunchKernelGGL((conv32d_kernel<KIND, SK, KSV, HEADV>), grid, block,        \
                     e->lds_bytes_d, e->stream, a)
#define FFN_E_LAUNCH(HEADV)                                                      \
  hipLaunchKernelGGL(                                                            \
      (conv32d_kernel<KIND, SK, kEPieces, HEADV, kETiles, kERows, 2>), grid,     \
      block, e->lds_bytes_e, e->stream, a)
  if (e->t_now) {
    // one FoV: 32-voxel tail workgroups (balance over the CUs); several: the
    // same voxels in 96-voxel ones (cost per voxel) -- the same bits
    const bool one = n == 1;
    ConvTailMap mp;
    tail_map(e, n, mp);
    const dim3 tgrid(8 * n * (mp.mains_per_xcd + mp.tails_per_xcd));
    const bool flow1 = one && e->flow == 1;
    if (flow1) {
      a.L.flow_wait_on = layer > 0;
      a.L.flow_wait = e->flow_epoch;
      a.L.flow_set = ++e->flow_epoch;
    }
#define FFN_MT_LAUNCH(HEADV)                                                      \
  if (flow1) {                                                                    \
    if constexpr (KIND == 0 || SK || !HEADV)                                      \
      hipLaunchKernelGGL((conv32mt_kernel<KIND, SK, HEADV, 1, true>), tgrid,      \
                         block, kMLdsBytes, e->stream, a, mp);                    \
  } else if (one)                                                                 \
    hipLaunchKernelGGL((conv32mt_kernel<KIND, SK, HEADV, 1>), tgrid, block,       \
                       kMLdsBytes, e->stream, a, mp);                             \
  else                                                                            \
    hipLaunchKernelGGL((conv32mt_kernel<KIND, SK, HEADV, 3>), tgrid, block,       \
                       kMLdsBytes, e->stream, a, mp)
    if (head.on) {
      if constexpr (KIND == 1) { FFN_MT_LAUNCH(true); }
    } else {
      FFN_MT_LAUNCH(false);
    }
#undef FFN_MT_LAUNCH
  } else if (msplit) {
    if (head.on) {
      if constexpr (KIND == 1)
        hipLaunchKernelGGL((conv32m_kernel<KIND, SK, true>), grid, block,
                           kMLdsBytes, e->stream, a);
    } else {
      hipLaunchKernelGGL((conv32m_kernel<KIND, SK, false>), grid, block,
                         kMLdsBytes, e->stream, a);
    }
  } else if (small) {
    if (head.on) {
      if constexpr (KIND == 1) FFN_E_LAUNCH(true);
    } else {
      FFN_E_LAUNCH(false);
    }
  } else if (head.on) {
    if constexpr (KIND == 1) {
      if (ks == 8) FFN_D_LAUNCH(8, true);
      else if (ks == 9) FFN_D_LAUNCH(9, true);
      else FFN_D_LAUNCH(10, true);
    }
  } else {
    if (ks == 8) FFN_D_LAUNCH(8, false);
    else if (ks == 9) FFN_D_LAUNCH(9, false);
    else FFN_D_LAUNCH(10, false);
  }
#undef FFN_D_LAUNCH
#undef FFN_E_LAUNCH
  if (prof) HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  return FFN_OK;
}

// FoVs described by `si` -> logits (+ count of logits >= move_thr, + seed_raw)
// conv0_a of a step whose range tag is `tag` (sp.n > 0: a speculative launch)
// conv0_a of the split-product family as the argument block of the fused launch;
// returns its number of tiles.  next: for the step AFTER the current one (its
// outputs go to the other set: ffn_engine::seed_raw_alt ...)
int conv0a_split_args(ffn_engine* e, float pad_value, unsigned tag, const SpecArgs& sp,
                      bool next, Conv0Next& nx) {
  const float* W = e->weights;
  const Geom& q = e->gp;  // the split-product kernels' layout of the FoV
  const int qz = (q.fz + kC0Z - 1) / kC0Z, qy = (q.fy + kC0Y - 1) / kC0Y,
            qx = (q.fx + kC0X - 1) / kC0X;
  nx.pad_value = pad_value;
  nx.w = W + (e->permuted ? e->w0ap_off : e->w0a_off);
  nx.bias = W + e->b0a_off;
  nx.out = e->bufT;
  nx.seed_raw = next ? e->seed_raw_alt : e->seed_raw;
  nx.q = q;
  nx.tiles_y = qy;
  nx.tiles_x = qx;
  nx.so.out_sp = reinterpret_cast<char*>(e->rawT) + (size_t)q.guard * 16;
  nx.so.sp_plane_bytes = (q.act_stride / kFeatures) * 16;
  nx.so.item_bytes = q.act_stride * (long)sizeof(float);
  nx.so.range_flag = next ? e->range_flag_alt : e->range_flag;
  nx.so.range_tag = tag;
  nx.sp = sp;
  if (sp.n > 0) nx.sp.choice = next ? e->d_spec_choice_alt : e->d_spec_choice;
  return qz * qy * qx;
}

void launch_conv0a(ffn_engine* e, int n, const StepItems& si, float pad_value,
                   unsigned tag, const SpecArgs& sp, bool next = false) {
  const Geom& g = e->g;
  const float* W = e->weights;
  const int tz = (g.fz + kC0Z - 1) / kC0Z, ty = (g.fy + kC0Y - 1) / kC0Y,
            tx = (g.fx + kC0X - 1) / kC0X;
  if (e->conv_variant >= 6) {
    Conv0Next nx;
    const int tiles = conv0a_split_args(e, pad_value, tag, sp, next, nx);
    hipLaunchKernelGGL(conv0a_mfma_kernel<true>, dim3(tiles, n),
                       dim3(kC0Threads), 0, e->stream, si, pad_value, nx.w, nx.bias,
                       nx.out, nx.seed_raw, nx.q, nx.tiles_y, nx.tiles_x, nx.so, nx.sp);
  } else {
    SpecArgs sp2 = sp;
    if (sp.n > 0) sp2.choice = next ? e->d_spec_choice_alt : e->d_spec_choice;
    hipLaunchKernelGGL(conv0a_mfma_kernel<false>, dim3(tz * ty * tx, n),
                       dim3(kC0Threads), 0, e->stream, si, pad_value,
                       W + e->w0a_off, W + e->b0a_off, e->bufT,
                       next ? e->seed_raw_alt : e->seed_raw, g, ty, tx,
                       Conv0SplitOut(), sp2);
  }
}

// The 2 depth - 1 convs of ONE FoV as a single resident launch (conv32ps,
// ffn_conv_resident.h): conv32mt's workgroups keep their voxels through the stack and
// hand rows to each other through the tile words instead of kernel boundaries.
// ev0 / ev1: a sampled launch -- the events carry the START and END of this very dispatch
// (hipExtLaunchKernelGGL: the timestamps rocprofv3 reads), not markers queued around it: a
// marker is a barrier packet with system-scope fences of its own, 2 - 3 us each, which the
// sampled launch and the step around it would wait behind
int launch_conv32ps(ffn_engine* e, float pad_value, float move_thr,
                    const int* ahead_choice = nullptr, hipEvent_t ev0 = nullptr,
                    hipEvent_t ev1 = nullptr) {
  HeadFusion hf;
  hf.on = true;
  hf.pad_value = pad_value;
  hf.move_thr = move_thr;
  ConvDArgs a;
  conv32d_args(e, 1, e->rawT, e->rawS, 0, hf, a);
  a.L.dbg = e->dbg_clock ? e->d_dbg : nullptr;
  ConvTailMap mp;
  tail_map(e, 1, mp);
  const Geom& g = e->gp;
  ConvStackTab tb;
  tb.nlayers = 2 * e->depth - 1;
  tb.dbg_layer = e->dbg_layer;
  tb.sp_t = reinterpret_cast<const char*>(e->rawT) + (size_t)g.guard * 16;
  tb.sp_s = reinterpret_cast<char*>(e->rawS) + (size_t)g.guard * 16;
  tb.wpack0 = reinterpret_cast<const char*>(e->wpackd);
  tb.wpack_stride = (long)(e->wpackd_layer * sizeof(uint16_t));
  tb.bias0 = e->weights + e->bias_off[0];
  tb.bias_stride = e->depth > 1 ? (long)(e->bias_off[1] - e->bias_off[0]) : 0;
  tb.epoch0 = e->flow_epoch;
  tb.pace = e->flow_pace < 0 ? e->pace_auto : e->flow_pace;
  tb.pace_tail = e->flow_pace_tail;
  tb.pace_spread = e->flow_pace_spread < 0 ? tb.pace : e->flow_pace_spread;
  e->flow_epoch += (unsigned)tb.nlayers;
  const dim3 grid(8 * (mp.mains_per_xcd + mp.tails_per_xcd)), block(kDThreads);
  tb.l_begin = 0;
  tb.l_end = tb.nlayers;
  tb.stamps = e->d_stamps;
  tb.ahead_choice = ahead_choice;
  // (2: the stack in front of the traced step, queued ahead one call earlier: its end only)
  tb.trace = e->trace_now ? 1 : (ahead_choice && e->fused_trace_in == 1) ? 2 : 0;
  const long long t_l0 = e->t_arrived_ns ? steady_ns() : 0;
  if (ev0)
    hipExtLaunchKernelGGL(conv32ps_kernel, grid, block, kMLdsBytes, e->stream, ev0, ev1, 0, a,
                          mp, tb);
  else
    hipLaunchKernelGGL(conv32ps_kernel, grid, block, kMLdsBytes, e->stream, a, mp, tb);
  if (e->t_arrived_ns) {
    const long long t_l1 = steady_ns();
    if (t_l1 - e->t_arrived_ns < 100000) {  // (inside a segment: see ConvStackTab::stamps)
      e->stat_turn_host_ns += t_l1 - e->t_arrived_ns;
      e->stat_launch_host_ns += t_l1 - t_l0;
      e->stat_turn_host_count += 1;
    }
    e->t_arrived_ns = 0;
  }
  return FFN_OK;
}

// The same stack on 64-voxel workgroups, two per CU (conv32hs, ffn_conv_half.h).
void half_map(const ffn_engine* e, ConvHalfMap& mp) {
  // (the dispatcher hands an XCD's first blocks to its empty CUs: cus / 8 first slots)
  mp.n_chunks = e->n_half;
  mp.per_first = std::max(1, e->cus / 8);
  mp.n_first = std::min(e->n_half, 8 * mp.per_first);
  mp.per_second = (e->n_half - mp.n_first + 7) / 8;
}

// one conv of the 80-voxel family as its own launch: what a resident step that came
// back void is repeated with (the same bits), and flow = 0
template <int KIND, bool SK>
int launch_conv32h(ffn_engine* e, const float* raw_in, float* raw_out, int layer,
                   const HeadFusion& head = HeadFusion()) {
  ConvDArgs a;
  conv32d_args(e, 1, raw_in, raw_out, layer, head, a);
  a.L.wpack = reinterpret_cast<const char*>(e->wpackh + (size_t)layer * e->wpackd_layer);
  a.flow_n_main = -1;
  ConvHalfMap mp;
  half_map(e, mp);
  const dim3 grid(8 * (mp.per_first + mp.per_second)), block(kDThreads);
  if (head.on) {
    if constexpr (KIND == 1)
      hipLaunchKernelGGL((conv32h_kernel<KIND, SK, true>), grid, block, kHLdsBytes, e->stream,
                         a, mp);
  } else {
    hipLaunchKernelGGL((conv32h_kernel<KIND, SK, false>), grid, block, kHLdsBytes, e->stream,
                       a, mp);
  }
  return FFN_OK;
}

int launch_conv32hs(ffn_engine* e, float pad_value, float move_thr) {
  HeadFusion hf;
  hf.on = true;
  hf.pad_value = pad_value;
  hf.move_thr = move_thr;
  ConvDArgs a;
  conv32d_args(e, 1, e->rawT, e->rawS, 0, hf, a);
  a.flow_n_main = -1;
  ConvHalfMap mp;
  half_map(e, mp);
  const Geom& g = e->gp;
  ConvStackTab tb;
  tb.nlayers = 2 * e->depth - 1;
  tb.dbg_layer = e->dbg_layer;
  tb.sp_t = reinterpret_cast<const char*>(e->rawT) + (size_t)g.guard * 16;
  tb.sp_s = reinterpret_cast<char*>(e->rawS) + (size_t)g.guard * 16;
  tb.wpack0 = reinterpret_cast<const char*>(e->wpackh);
  tb.wpack_stride = (long)(e->wpackd_layer * sizeof(uint16_t));
  tb.bias0 = e->weights + e->bias_off[0];
  tb.bias_stride = e->depth > 1 ? (long)(e->bias_off[1] - e->bias_off[0]) : 0;
  tb.epoch0 = e->flow_epoch;
  e->flow_epoch += (unsigned)tb.nlayers;
  tb.l_begin = 0;
  tb.l_end = tb.nlayers;
  tb.pace = e->flow_pace < 0 ? 0 : e->flow_pace;  // (no measured beat for this family)
  tb.pace_tail = 0;
  tb.pace_spread = e->flow_pace_spread < 0 ? tb.pace : e->flow_pace_spread;
  tb.stamps = nullptr;
  tb.ahead_choice = nullptr;
  tb.trace = 0;
  const dim3 grid(8 * (mp.per_first + mp.per_second)), block(kDThreads);
  hipLaunchKernelGGL(conv32hs_kernel, grid, block, kHLdsBytes, e->stream, a, mp, tb);
  return FFN_OK;
}

// conv0a_done: the step's conv0_a has been queued already (a speculative launch
// that chose its position)
// ahead: the stack of the step AFTER the one being submitted, behind the speculative
// conv0_a just queued for it (engine option stack_ahead; conv0a_done with it)
int run_stack(ffn_engine* e, int n, const StepItems& si, float pad_value,
              float move_thr, bool conv0a_done = false, bool ahead = false) {
  const Geom& g = e->g;
  const float* W = e->weights;
  if (!conv0a_done) drop_spec(e);
  if (e->ahead_valid) e->stat_ahead_wasted += 1;  // (a caller that steps past it)
  e->spec.valid = false;
  e->ahead_valid = false;
  e->last_stack_resident = false;
  e->range_tag = next_tag(e->range_tag);
  // this step's set of conv0_a outputs: what a launch made ahead for it wrote,
  // what its own conv0_a (below) writes
  std::swap(e->seed_raw, e->seed_raw_alt);
  std::swap(e->range_flag, e->range_flag_alt);
  std::swap(e->d_spec_choice, e->d_spec_choice_alt);
  const bool sampled = (e->stack_calls % e->prof_every) == 0;
  e->stack_calls++;
  e->prof_now = e->prof_mode == 1 && sampled;
  const bool prof_chain = e->prof_mode == 2 && sampled;
  if (!conv0a_done) launch_conv0a(e, n, si, pad_value, e->range_tag, SpecArgs());
  int rc;
  const float* head_in;
  bool head_fused = false;
  e->ahead_ev_pending = false;
  // a sampled stack that runs as ONE resident launch is timed by that dispatch's own
  // start / end (launch_conv32ps); a chain of launches by markers around it
  const bool one_launch = e->conv_variant == 9 && n == 1 && e->flow == 2 &&
                          e->flow_skip == 0 && e->tail_batched == 0;
  hipEvent_t k_ev0 = nullptr, k_ev1 = nullptr;
  if (prof_chain && ahead) {
    if (one_launch) k_ev0 = e->ahead_ev[0], k_ev1 = e->ahead_ev[1];
    else HIP_TRY(hipEventRecord(e->ahead_ev[0], e->stream));
  } else if (prof_chain) {
    if (e->events_used + 2 > (int)e->events.size()) {
      rc = flush_events(e);
      if (rc) return rc;
    }
    e->chain_launches_pending.push_back(2 * e->depth - 1);
    if (one_launch) {
      k_ev0 = e->events[e->events_used], k_ev1 = e->events[e->events_used + 1];
      e->events_used += 2;
    } else {
      HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
    }
  }
  // conv32d: one workgroup per CU (160-voxel chunks) for a single FoV; with
  // several FoVs in flight the 96-voxel form (two workgroups per CU, the same
  // arithmetic bit for bit) overlaps one workgroup's MFMAs with the other's
  // staging and epilogue
  e->small_now = e->conv_variant == 7 ||
                 (e->conv_variant == 6 && e->batch_chunks == 1 && n >= 2 && e->e_ok);
  e->m_now = e->conv_variant >= 8 ||
             (e->conv_variant == 6 && e->batch_chunks == 2 && n >= 2 && e->m_ok);
  // variant 9: a single FoV runs conv32mt (balance over the CUs: +14 %); steps
  // with several FoVs run plain conv32m (cost per voxel: the K-split tail costs
  // 5-10 % there) unless tail_batched asks for the single-FoV bits
  e->t_now = e->conv_variant == 9 && (n == 1 || e->tail_batched != 0);
  // variant 10: a single FoV on 80-voxel workgroups (conv32hs / conv32h); steps with
  // several FoVs run plain conv32m, as under 9
  e->h_now = e->conv_variant == 10 && n == 1;
  if (e->conv_variant >= 6) {
    if (e->depth == 1)
      return fail(FFN_ERR_ARG, "conv_variant 6 needs depth >= 2 (fused head)");
    // T' -> (X, X') -> T' -> ... ; the head is always fused into the last conv_b
    auto chain = [&]() -> int {
      if (e->h_now) {
        if (e->flow == 2 && e->flow_skip > 0) {
          e->flow_skip -= 1;  // the repeat of a voided resident step
        } else if (e->flow == 2) {
          e->last_stack_resident = true;
          return launch_conv32hs(e, pad_value, move_thr);
        }
        int r = launch_conv32h<1, false>(e, e->rawT, e->rawS, 0);
        for (int i = 1; i < e->depth && !r; ++i) {
          r = launch_conv32h<0, false>(e, e->rawS, e->rawT, 2 * i - 1);
          if (r) break;
          HeadFusion hf;
          hf.on = i == e->depth - 1;
          hf.pad_value = pad_value;
          hf.move_thr = move_thr;
          r = launch_conv32h<1, true>(e, e->rawT, e->rawS, 2 * i, hf);
        }
        return r;
      }
      if (e->t_now && n == 1 && e->flow == 2) {
        if (e->flow_skip > 0) {
          e->flow_skip -= 1;  // the repeat of a voided resident step
        } else {
          e->last_stack_resident = true;
          return launch_conv32ps(e, pad_value, move_thr, ahead ? e->d_spec_choice : nullptr,
                                 k_ev0, k_ev1);
        }
      }
      if (ahead) return fail(FFN_ERR_STATE, "stack_ahead without the resident launch");
      int r = launch_conv32d<1, false>(e, n, e->rawT, e->rawS, 0);
      for (int i = 1; i < e->depth && !r; ++i) {
        r = launch_conv32d<0, false>(e, n, e->rawS, e->rawT, 2 * i - 1);
        if (r) break;
        HeadFusion hf;
        hf.on = i == e->depth - 1;
        hf.pad_value = pad_value;
        hf.move_thr = move_thr;
        r = launch_conv32d<1, true>(e, n, e->rawT, e->rawS, 2 * i, hf);
      }
      return r;
    };
    rc = chain();
    if (rc) return rc;
    head_fused = true;
    head_in = e->bufX;
  } else if (e->conv_variant == 0) {
    rc = launch_conv32<false, false, false>(e, n, e->bufT, e->bufX, nullptr, 0);
    if (rc) return rc;
    for (int i = 1; i < e->depth; ++i) {
      rc = launch_conv32<true, true, false>(e, n, e->bufX, e->bufT, nullptr,
                                            2 * i - 1);
      if (rc) return rc;
      rc = launch_conv32<false, false, true>(e, n, e->bufT, e->bufX, e->bufX,
                                             2 * i);
      if (rc) return rc;
    }
    head_in = e->bufX;
  } else {
    // conv_variant 2: T is post-ReLU (conv0_a / conv_a apply it), X is the raw
    // residual stream
    rc = launch_conv32c<false, false, false>(e, n, e->bufT, e->bufX, nullptr, 0);
    if (rc) return rc;
    for (int i = 1; i < e->depth; ++i) {
      rc = launch_conv32c<true, true, false>(e, n, e->bufX, e->bufT, nullptr,
                                             2 * i - 1);
      if (rc) return rc;
      HeadFusion hf;
      hf.on = e->fuse_head && i == e->depth - 1 && e->ablate == 0;
      hf.pad_value = pad_value;
      hf.move_thr = move_thr;
      rc = launch_conv32c<false, false, true>(e, n, e->bufT, e->bufX, e->bufX,
                                              2 * i, hf);
      if (rc) return rc;
      head_fused = hf.on;
    }
    head_in = e->bufX;
  }
  if (prof_chain && ahead) {
    if (!k_ev0) HIP_TRY(hipEventRecord(e->ahead_ev[1], e->stream));
    e->ahead_ev_pending = true;
  } else if (prof_chain && !k_ev0) {
    HIP_TRY(hipEventRecord(e->events[e->events_used++], e->stream));
  }
  if (head_fused) {
    e->count_blocks = e->h_now ? e->n_half
                      : e->t_now ? e->n_main + (n == 1 ? e->n_tail : e->n_tail3)
                      : e->m_now ? e->nchunks_m : e->small_now ? e->nchunks_e
                      : e->conv_variant >= 6 ? e->nchunks_k : e->nchunks_c;
  } else {
    e->count_blocks = kHeadBlocks;
    hipLaunchKernelGGL(head_kernel, dim3(kHeadBlocks, n), dim3(256), 0, e->stream,
                       head_in, e->seed_raw, pad_value, W + e->wl_off, move_thr,
                       e->logits, e->count, g);
  }
  HIP_TRY(hipGetLastError());
  return FFN_OK;
}

// Step descriptors for the dense uploaded FoVs (predict / forward_resident):
// each FoV is its own FoV-sized "canvas" centred on its middle voxel.
int dense_items(ffn_engine* e, int n, StepItems* si) {
  const Geom& g = e->g;
  for (int k = 0; k < n; ++k) {
    StepItem& it = e->h_items[k];
    std::memset(&it, 0, sizeof(it));
    it.image = e->up_image + (size_t)k * g.V;
    it.image_u8 = nullptr;
    it.image_lut = nullptr;
    it.seed = e->up_seed + (size_t)k * g.V;
    it.seg = nullptr;
    it.cz = g.fz;
    it.cy = g.fy;
    it.cx = g.fx;
    it.req.pos[0] = g.fz / 2;
    it.req.pos[1] = g.fy / 2;
    it.req.pos[2] = g.fx / 2;
  }
  si->items = e->d_items;
  si->use_inline = n == 1;
  si->inline_item = e->h_items[0];
  if (n > 1)
    HIP_TRY(hipMemcpyAsync(e->d_items, e->h_items, sizeof(StepItem) * n,
                           hipMemcpyHostToDevice, e->stream));
  return FFN_OK;
}

Box make_box(const ffn_canvas* c, const int32_t lo[3], const int32_t hi[3],
             long* total) {
  Box b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = lo[k];
    b.n[k] = hi[k] - lo[k];
  }
  b.cy = c->cy;
  b.cx = c->cx;
  *total = (long)b.n[0] * b.n[1] * b.n[2];
  return b;
}

int check_canvas(const ffn_canvas* c) {
  if (!c) return fail(FFN_ERR_ARG, "null canvas");
  if (!c->engine) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  return FFN_OK;
}

int check_box(const ffn_canvas* c, const int32_t lo[3], const int32_t hi[3]) {
  const int dims[3] = {c->cz, c->cy, c->cx};
  for (int k = 0; k < 3; ++k)
    if (lo[k] < 0 || hi[k] > dims[k] || hi[k] < lo[k])
      return fail(FFN_ERR_ARG, "box [%d,%d) out of canvas axis %d (size %d)",
                  lo[k], hi[k], k, dims[k]);
  return FFN_OK;
}

int grid_for(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)std::max<long>(1, std::min<long>(g, 2048));
}

}  // namespace

extern "C" {

int ffn_abi_version(void) { return 10; }

const char* ffn_last_error(void) { return g_error.c_str(); }

size_t ffn_engine_weight_count(int depth, int features) {
  const size_t F = (size_t)features;
  return 27 * 2 * F + F + (size_t)(2 * depth - 1) * (27 * F * F + F) + F + 1;
}

int ffn_engine_create(int device_id, const int32_t fov_zyx[3],
                      const int32_t deltas_zyx[3], int depth, int features,
                      int max_batch, ffn_engine** out) {
  if (!out || !fov_zyx || !deltas_zyx) return fail(FFN_ERR_ARG, "null argument");
  *out = nullptr;
  if (features != kFeatures)
    return fail(FFN_ERR_ARG, "features must be %d (got %d)", kFeatures, features);
  if (depth < 1 || max_batch < 1)
    return fail(FFN_ERR_ARG, "depth and max_batch must be >= 1");
  for (int k = 0; k < 3; ++k) {
    if (fov_zyx[k] < 3 || fov_zyx[k] % 2 == 0)
      return fail(FFN_ERR_ARG, "fov must be odd and >= 3 per axis");
    if (deltas_zyx[k] < 0 || 2 * deltas_zyx[k] + 1 > fov_zyx[k])
      return fail(FFN_ERR_ARG, "deltas do not fit inside the fov");
  }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev)
    return fail(FFN_ERR_ARG, "device %d not present (%d visible)", device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));

  ffn_engine* e = new ffn_engine();
  e->device = device_id;
  e->depth = depth;
  e->max_batch = max_batch;
  Geom& g = e->g;
  g.fz = fov_zyx[0];
  g.fy = fov_zyx[1];
  g.fx = fov_zyx[2];
  g.dz = deltas_zyx[0];
  g.dy = deltas_zyx[1];
  g.dx = deltas_zyx[2];
  g.XS = g.fx + 1;
  g.plane = (g.fy + 1) * g.XS;
  g.npos = g.fz * g.plane;
  g.guard = g.plane + g.XS + 1;
  g.nchunks = (g.npos + kChunk - 1) / kChunk;
  g.V = g.fz * g.fy * g.fx;
  g.R = kChunk + 2 * (g.XS + 1);
  g.oa[0] = 0, g.oa[1] = 1, g.oa[2] = 2;
  g.dstr[0] = g.fy * g.fx, g.dstr[1] = g.fx, g.dstr[2] = 1;
  g.crop = 0;
  g.c0[0] = g.c0[1] = g.c0[2] = 0;
  g.c1[0] = g.fz, g.c1[1] = g.fy, g.c1[2] = g.fx;
  g.Vp = g.V;
  // the geometry with its axes permuted: internal axis a = original axis oa[a]
  auto permute = [&](const int oa[3]) {
    const int f[3] = {g.fz, g.fy, g.fx}, d[3] = {g.dz, g.dy, g.dx};
    const int ds[3] = {g.fy * g.fx, g.fx, 1};
    Geom q = g;
    q.fz = f[oa[0]], q.fy = f[oa[1]], q.fx = f[oa[2]];
    q.dz = d[oa[0]], q.dy = d[oa[1]], q.dx = d[oa[2]];
    q.XS = q.fx + 1;
    q.plane = (q.fy + 1) * q.XS;
    q.npos = q.fz * q.plane;
    q.guard = q.plane + q.XS + 1;
    q.nchunks = (q.npos + kChunk - 1) / kChunk;
    q.R = kChunk + 2 * (q.XS + 1);
    for (int a = 0; a < 3; ++a) q.oa[a] = oa[a], q.dstr[a] = ds[oa[a]];
    return q;
  };
  // largest padded span of `count` chunks of `chunk` dense voxels from `start`
  auto chunk_span = [](const Geom& q, int start, int chunk, int count) {
    auto pad = [&](int v) {
      const int x = v % q.fx, y = (v / q.fx) % q.fy, z = v / (q.fx * q.fy);
      return z * q.plane + y * q.XS + x;
    };
    int span = 0;
    for (int c = 0; c < count; ++c) {
      const int lo = start + c * chunk, hi = std::min(q.V, lo + chunk) - 1;
      if (lo <= hi) span = std::max(span, pad(hi) - pad(lo) + 1);
    }
    return span;
  };
  auto m_fits = [&](const Geom& q) {  // conv32m's 128-voxel chunks in kMRows rows
    return chunk_span(q, 0, kMChunk, (q.V + kMChunk - 1) / kMChunk) +
               2 * (q.XS + 1) <= kMRows;
  };
  e->gp = g;
  if (!m_fits(g)) {
    // shortest axis last (= the row direction), the other two in their order
    static const int kPerms[5][3] = {{0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1},
                                     {2, 1, 0}};
    int best = -1;
    for (int k = 0; k < 5; ++k) {
      const Geom q = permute(kPerms[k]);
      if (m_fits(q) && (best < 0 || q.fx < permute(kPerms[best]).fx)) best = k;
    }
    if (best >= 0) {
      e->gp = permute(kPerms[best]);
      e->permuted = true;
    }
  }
  const long positions = std::max(
      (long)g.guard + (long)g.nchunks * kChunk + g.guard + 320,
      (long)e->gp.guard + (long)e->gp.nchunks * kChunk + e->gp.guard + 320);
  g.act_stride = positions * kFeatures;
  e->gp.act_stride = g.act_stride;
  e->lds_bytes = (size_t)3 * g.R * kFeatures * sizeof(float);
  if (e->lds_bytes > 160 * 1024) {
    delete e;
    return fail(FFN_ERR_ARG, "fov too wide for the LDS-staged conv (%zu B)",
                e->lds_bytes);
  }

#define E_TRY(expr)                                                          \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      int _rc = fail(FFN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                     hipGetErrorString(_e), __FILE__, __LINE__);             \
      ffn_engine_destroy(e);                                                 \
      return _rc;                                                            \
    }                                                                        \
  } while (0)

  E_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  {
    int lo = 0, hi = 0;  // (numerically lowest = highest priority)
    E_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    E_TRY(hipStreamCreateWithPriority(&e->ustream, hipStreamNonBlocking, hi));
  }
  E_TRY(hipEventCreateWithFlags(&e->main_ev, hipEventDisableTiming));
  const size_t act_bytes = (size_t)3 * max_batch * g.act_stride * sizeof(float);
  E_TRY(hipMalloc(&e->act_base, act_bytes));
  E_TRY(hipMemset(e->act_base, 0, act_bytes));
  e->rawT = e->act_base;
  e->rawX = e->rawT + (size_t)max_batch * g.act_stride;
  e->rawS = e->rawX + (size_t)max_batch * g.act_stride;
  e->bufT = e->rawT + (size_t)g.guard * kFeatures;
  e->bufX = e->rawX + (size_t)g.guard * kFeatures;
  const size_t vbytes = (size_t)max_batch * g.V * sizeof(float);
  E_TRY(hipMalloc(&e->up_image, vbytes));
  E_TRY(hipMalloc(&e->up_seed, vbytes));
  E_TRY(hipMalloc(&e->seed_raw, vbytes));
  E_TRY(hipMalloc(&e->seed_raw_alt, vbytes));
  E_TRY(hipMalloc(&e->logits, vbytes));
  E_TRY(hipHostMalloc(&e->h_io, 3 * vbytes, hipHostMallocDefault));
  E_TRY(hipMemset(e->up_image, 0, vbytes));
  E_TRY(hipMemset(e->up_seed, 0, vbytes));
  E_TRY(hipMalloc(&e->count, sizeof(unsigned) * max_batch *
                                  std::max<size_t>(kHeadBlocks, (g.V + 31) / 32)));
  E_TRY(hipMalloc(&e->d_items, sizeof(StepItem) * 2 * max_batch));
  E_TRY(hipHostMalloc(&e->h_items, sizeof(StepItem) * 2 * max_batch,
                      hipHostMallocDefault));
  E_TRY(hipHostMalloc(&e->h_pub,
                      sizeof(unsigned long long) * kPubWords * 2 * max_batch,
                      hipHostMallocDefault));
  std::memset(e->h_pub, 0, sizeof(unsigned long long) * kPubWords * 2 * max_batch);

  // validity table of the padded-flat layout
  {
    std::vector<uint8_t> v((size_t)g.nchunks * kChunk, 0);
    for (int p = 0; p < g.npos; ++p) {
      const int rem = p % g.plane;
      v[p] = (rem / g.XS < g.fy && rem % g.XS < g.fx) ? 1 : 0;
    }
    E_TRY(hipMalloc(&e->valid, v.size()));
    E_TRY(hipMemcpy(e->valid, v.data(), v.size(), hipMemcpyHostToDevice));
    std::vector<uint32_t> bits((size_t)g.nchunks * 5, 0u);
    for (size_t p = 0; p < v.size(); ++p)
      if (v[p]) bits[(p / kChunk) * 5 + ((p % kChunk) >> 5)] |= 1u << (p & 31);
    static_assert(kChunk == 160, "validbits layout assumes 5 words per chunk");
    E_TRY(hipMalloc(&e->validbits, bits.size() * sizeof(uint32_t)));
    E_TRY(hipMemcpy(e->validbits, bits.data(), bits.size() * sizeof(uint32_t),
                    hipMemcpyHostToDevice));
  }

  // dense -> padded position table and LDS extent of the compact variant
  {
    e->nchunks_c = (g.V + kCChunk - 1) / kCChunk;
    e->nchunks_k = (g.V + kDChunk - 1) / kDChunk;
    std::vector<int32_t> pidx(std::max((size_t)e->nchunks_c * kCChunk,
                                       (size_t)e->nchunks_k * kDChunk));
    for (size_t v = 0; v < pidx.size(); ++v) {
      const int vv = (int)std::min<size_t>(v, (size_t)g.V - 1);
      const int x = vv % g.fx, y = (vv / g.fx) % g.fy, z = vv / (g.fx * g.fy);
      pidx[v] = z * g.plane + y * g.XS + x;
    }
    int span = 0;
    for (int c = 0; c < e->nchunks_c; ++c)
      span = std::max(span, pidx[(size_t)c * kCChunk + kCChunk - 1] -
                                pidx[(size_t)c * kCChunk] + 1);
    e->Rc = ((span + 2 * (g.XS + 1)) + 31) / 32 * 32;
    if (e->Rc < 256) e->Rc = 256;  // the kernel stages 8 or 9 x 256 float4
    e->lds_bytes_c = (size_t)2 * e->Rc * kCLdsStride * sizeof(float);
    {
      // everything from here on is the split-product family: laid out as gp
      const Geom& q = e->gp;
      const int span_k = chunk_span(q, 0, kDChunk, e->nchunks_k);
      e->Rc_k = ((span_k + 2 * (q.XS + 1)) + 31) / 32 * 32;
      if (e->Rc_k < 256) e->Rc_k = 256;
      e->k_ok = e->Rc_k <= 320 && e->nchunks_k >= 2;  // (magic divisions: d >= 2)
      // tap schedule of the four waves (see conv32d_body): two dz = -1 taps
      // each, then two taps of dz <= 0, then the rest; 7 / 7 / 7 / 6 taps
      static const int kSched[4][7] = {{0, 1, 8, 9, 16, 18, 19},
                                       {2, 3, 10, 11, 17, 20, 21},
                                       {4, 5, 12, 13, 22, 23, 24},
                                       {6, 7, 14, 15, 25, 26, -1}};
      // in the plane-major LDS image (8 planes x Rc_k
      // rows x 16 B per segment); wave 3's seventh tap is the all-zero tap 27
      e->lds_bytes_d = std::max((size_t)3 * 128 * e->Rc_k,
                                (size_t)4 * kDChunk * kDRowB + 64);
      for (int w = 0; w < 4; ++w)
        for (int j = 0; j < 7; ++j) {
          int s = kSched[w][j];
          const bool dummy = s < 0;
          if (dummy) s = kSched[w][j - 1];
          const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
          e->dsched_aoff[w * 8 + j] =
              kz * 128 * e->Rc_k + ((ky - 1) * q.XS + (kx - 1)) * 16;
          e->dsched_btap[w * 8 + j] = dummy ? 27 : s;
        }
      e->d_ok = e->k_ok && e->lds_bytes_d <= 160 * 1024 && depth >= 2;
      // variant 7: 96-voxel chunks in kERows rows, three slots in 80 KB
      {
        const int ce = 32 * kETiles;
        e->nchunks_e = (q.V + ce - 1) / ce;
        const int span_e = chunk_span(q, 0, ce, e->nchunks_e);
        e->lds_bytes_e = std::max((size_t)3 * 128 * kERows,
                                  (size_t)4 * ce * kDRowB + 64);
        e->e_ok = e->d_ok && span_e + 2 * (q.XS + 1) <= kERows &&
                  e->nchunks_e >= 2 && e->lds_bytes_e <= 80 * 1024;
        // variant 8: 128-voxel chunks in kMRows rows
        e->nchunks_m = (q.V + kMChunk - 1) / kMChunk;
        const int span_m = chunk_span(q, 0, kMChunk, e->nchunks_m);
        e->m_ok = e->d_ok && span_m + 2 * (q.XS + 1) <= kMRows && e->nchunks_m >= 2;
        // variant 9: the chunks past the 256th become 32-voxel tail workgroups,
        // as long as all of one FoV's workgroups find a slot at once (two per CU)
        if (e->m_ok && e->nchunks_m > 256) {
          e->n_main = 256;
          e->n_tail = (q.V - 256 * kMChunk + 31) / 32;
          e->n_tail3 = (q.V - 256 * kMChunk + 95) / 96;
          const int span_t = chunk_span(q, 256 * kMChunk, 32, e->n_tail);
          const int span_t3 = chunk_span(q, 256 * kMChunk, 96, e->n_tail3);
          static_assert((size_t)3 * 128 * kTRows <= kMLdsBytes &&
                            (size_t)4 * 32 * kDRowB + 64 <= kMLdsBytes &&
                            (size_t)3 * 128 * kT3Rows <= kMLdsBytes &&
                            (size_t)4 * 96 * kDRowB + 64 <= kMLdsBytes,
                        "a tail workgroup fits conv32m's LDS");
          e->t_ok = span_t + 2 * (q.XS + 1) <= kTRows &&
                    span_t3 + 2 * (q.XS + 1) <= kT3Rows && e->n_tail <= 256;
          for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 7; ++j) {
              int s = kSched[w][j];
              if (s < 0) s = kSched[w][j - 1];
              const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
              e->tsched_aoff[w * 8 + j] =
                  kz * 128 * kTRows + ((ky - 1) * q.XS + (kx - 1)) * 16;
              e->t3sched_aoff[w * 8 + j] =
                  kz * 128 * kT3Rows + ((ky - 1) * q.XS + (kx - 1)) * 16;
            }
        }
        for (int w = 0; w < 4; ++w)
          for (int j = 0; j < 7; ++j) {
            int s = kSched[w][j];
            if (s < 0) s = kSched[w][j - 1];
            const int kz = s / 9, ky = (s / 3) % 3, kx = s % 3;
            e->esched_aoff[w * 8 + j] =
                kz * 128 * kERows + ((ky - 1) * q.XS + (kx - 1)) * 16;
          }
      }
    }
    E_TRY(hipMalloc(&e->d_dbg, (24 + 4 * kDbgMaxWgs) * sizeof(long long)));
    E_TRY(hipMemset(e->d_dbg, 0, (24 + 4 * kDbgMaxWgs) * sizeof(long long)));
    E_TRY(hipMalloc(&e->pidx, pidx.size() * sizeof(int32_t)));
    E_TRY(hipMemcpy(e->pidx, pidx.data(), pidx.size() * sizeof(int32_t),
                    hipMemcpyHostToDevice));
    // variant 2 needs Rc in {256, 288}
    const bool c_ok = e->Rc == 256 || e->Rc == 288;
    // default: conv32mt / conv32m where the geometry allows them (33^3: yes), else
    // conv32d, else the exact-f32 kernels (conv32c, or conv32 for any FoV that
    // fits the LDS at all)
    e->exact_variant = c_ok ? 2 : 0;
    e->conv_variant = e->t_ok ? 9 : e->m_ok ? 8 : e->d_ok ? 6 : e->exact_variant;
    // a single-FoV step of conv32mt runs its convs as ONE resident launch --
    // where the device can hold all of its workgroups at once: they wait for
    // each other, so one that is not on the chip stalls the rest until their
    // polls give up (a partitioned device, fewer CUs than main chunks)
    if (e->t_ok && depth >= 2) {
      int cus = 0, per_cu = 0;
      E_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id));
      E_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv32ps_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kMLdsBytes));
      E_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv32ps_kernel,
                                                         kDThreads, kMLdsBytes));
      ConvTailMap mp;
      tail_map(e, 1, mp);
      const int grid = 8 * (mp.mains_per_xcd + mp.tails_per_xcd);
      e->flow_fits = cus >= e->n_main && (long)cus * per_cu >= grid;
    }
    e->flow = e->flow_fits ? 2 : 0;
    // conv32hs (variant 10): 80-voxel workgroups, two per CU, all resident at once
    if (e->t_ok && depth >= 2) {
      const Geom& q = e->gp;
      e->n_half = FFN_H_VCLIP ? FFN_H_VCLIP / kHChunk : (q.V + kHChunk - 1) / kHChunk;
      const int span_h = chunk_span(q, 0, kHChunk, e->n_half);
      int cus = 0, per_cu = 0;
      E_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id));
      e->cus = cus;
      E_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv32hs_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kHLdsBytes));
      E_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv32hs_kernel, kDThreads,
                                                         kHLdsBytes));
      ConvHalfMap mp;
      half_map(e, mp);
      e->h_ok = e->n_half >= 2 && span_h + 2 * (q.XS + 1) <= kHRows &&
                (long)cus * per_cu >= 8 * (mp.per_first + mp.per_second);
    }
  }

  // weights: [w0a 27*2*32][b0a 32] ([wpack 27*32*32][bias 32]) x (2*depth-1)
  // [wl 32 + 1]
  {
    size_t off = 0;
    e->w0a_off = off;
    off += 27 * 2 * kFeatures;
    e->b0a_off = off;
    off += kFeatures;
    e->w0ap_off = off;
    off += 27 * 2 * kFeatures;
    for (int l = 0; l < 2 * depth - 1; ++l) {
      e->wpack_off.push_back(off);
      off += 27 * kFeatures * kFeatures;
      e->bias_off.push_back(off);
      off += kFeatures;
    }
    e->wl_off = off;
    off += kFeatures + 1;
    off = (off + 3) & ~(size_t)3;
    E_TRY(hipMalloc(&e->weights, off * sizeof(float)));
    e->wpackd_layer = (size_t)kDTaps * 2 * 2 * 64 * 8;  // + the all-zero tap
    E_TRY(hipMalloc(&e->wpackd, e->wpackd_layer * (2 * depth - 1) *
                                    sizeof(uint16_t)));
    E_TRY(hipMalloc(&e->wpackh, e->wpackd_layer * (2 * depth - 1) * sizeof(uint16_t)));
    // two words each: [0] the tag of the last void run, [1] ... of the last run that
    // was void because the resident launch timed out (the cause, per step)
    E_TRY(hipMalloc(&e->range_flag, 2 * sizeof(unsigned)));
    E_TRY(hipMemset(e->range_flag, 0, 2 * sizeof(unsigned)));
    E_TRY(hipMalloc(&e->range_flag_alt, 2 * sizeof(unsigned)));
    E_TRY(hipMemset(e->range_flag_alt, 0, 2 * sizeof(unsigned)));
    {
      const size_t words = ((size_t)(e->gp.V + 31) / 32 + 64) * kFlowStride;
      E_TRY(hipMalloc(&e->flow_flags, words * sizeof(unsigned)));
      E_TRY(hipMemset(e->flow_flags, 0, words * sizeof(unsigned)));
      E_TRY(hipMalloc(&e->flow_err, sizeof(unsigned)));
      E_TRY(hipMemset(e->flow_err, 0, sizeof(unsigned)));
      if (e->t_ok) {
        e->flow_trace_slots = std::max(e->n_main + e->n_tail, (e->gp.V + kHChunk - 1) / kHChunk);
        const size_t bytes = (size_t)e->flow_trace_slots * kFlowTraceLayers * 8 * sizeof(long long);
        E_TRY(hipMalloc(&e->flow_trace, bytes));
        E_TRY(hipMemset(e->flow_trace, 0, bytes));
      }
    }
    E_TRY(hipMalloc(&e->d_stamps, (32 + 1024) * sizeof(long long)));
    E_TRY(hipMemset(e->d_stamps, 0, (32 + 1024) * sizeof(long long)));
    E_TRY(hipMalloc(&e->d_spec_choice, sizeof(int)));
    E_TRY(hipMemset(e->d_spec_choice, 0xff, sizeof(int)));
    E_TRY(hipMalloc(&e->d_spec_choice_alt, sizeof(int)));
    E_TRY(hipMemset(e->d_spec_choice_alt, 0xff, sizeof(int)));
  }

  e->events.resize(2 * 64);
  for (auto& ev : e->events) E_TRY(hipEventCreate(&ev));
  for (auto& ev : e->ahead_ev) E_TRY(hipEventCreate(&ev));

  {
    int rc = set_lds_attr<false, false, false>(e->lds_bytes);
    if (!rc) rc = set_lds_attr<true, true, false>(e->lds_bytes);
    if (!rc) rc = set_lds_attr<false, false, true>(e->lds_bytes);
    if (!rc) rc = set_lds_attr_c<false, false, false>(e->lds_bytes_c);
    if (!rc) rc = set_lds_attr_c<true, true, false>(e->lds_bytes_c);
    if (!rc) rc = set_lds_attr_c<false, false, true>(e->lds_bytes_c);
    if (!rc) rc = set_lds_attr_c<false, false, true, 8>(e->lds_bytes_c);
    if (!rc) rc = set_lds_attr_c<false, false, true, 16>(e->lds_bytes_c);
    if (!rc) rc = set_lds_attr_c<false, false, true, 24>(e->lds_bytes_c);
    if (!rc && e->d_ok) rc = set_lds_attr_d(e->lds_bytes_d);
    if (!rc && e->e_ok) rc = set_lds_attr_e(e->lds_bytes_e);
    if (!rc && e->m_ok) rc = set_lds_attr_m();
    if (rc) {
      ffn_engine_destroy(e);
      return rc;
    }
  }
#undef E_TRY
  *out = e;
  return FFN_OK;
}

void ffn_engine_destroy(ffn_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  // Orphan the canvases still alive: their device memory goes with the engine;
  // ffn_canvas_destroy on an orphan only frees the host struct.
  for (ffn_canvas* c : e->canvases) {
    (void)hipFree(c->image);
    (void)hipFree(c->image_u8);
    (void)hipFree(c->image_lut);
    (void)hipFree(c->seed);
    (void)hipFree(c->seg);
    if (c->ev_util) (void)hipEventDestroy(c->ev_util);
    c->ev_util = nullptr;
    c->image = c->seed = nullptr;
    c->image_u8 = nullptr;
    c->image_lut = nullptr;
    c->seg = nullptr;
    c->engine = nullptr;
  }
  e->canvases.clear();
  if (e->ustream) (void)hipStreamSynchronize(e->ustream);
  if (e->main_ev) (void)hipEventDestroy(e->main_ev);
  if (e->ustream) (void)hipStreamDestroy(e->ustream);
  for (auto& ev : e->events)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ahead_ev)
    if (ev) (void)hipEventDestroy(ev);
  (void)hipFree(e->act_base);
  (void)hipFree(e->up_image);
  (void)hipFree(e->up_seed);
  (void)hipFree(e->seed_raw);
  (void)hipFree(e->seed_raw_alt);
  (void)hipFree(e->range_flag_alt);
  (void)hipFree(e->d_spec_choice_alt);
  (void)hipFree(e->logits);
  (void)hipFree(e->count);
  (void)hipFree(e->wpackd);
  (void)hipFree(e->wpackh);
  (void)hipFree(e->range_flag);
  (void)hipFree(e->flow_flags);
  (void)hipFree(e->flow_err);
  (void)hipFree(e->flow_trace);
  (void)hipFree(e->d_spec_choice);
  (void)hipFree(e->d_stamps);
  (void)hipFree(e->valid);
  (void)hipFree(e->validbits);
  (void)hipFree(e->pidx);
  (void)hipFree(e->d_dbg);
  (void)hipFree(e->weights);
  (void)hipFree(e->d_items);
  (void)hipFree(e->d_scratch);
  if (e->h_io) (void)hipHostFree(e->h_io);
  if (e->h_items) (void)hipHostFree(e->h_items);
  if (e->h_pub) (void)hipHostFree(e->h_pub);
  if (e->h_scratch) (void)hipHostFree(e->h_scratch);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// The beat of the paced resident stack is measured, not assumed: the chain a conv of the
// stack has to get through (words seen -> rows staged -> taps -> stores drained -> word
// published) is 6.2 - 7.0 us depending on the box and its clocks; a beat below it leaves
// the stack free-running (no loss), a beat above it costs 24 x the excess -- and on a box
// whose free-running stack already runs at its chain's length pacing gains nothing at all
// (profiles/r06_pacing.txt: -5 % on three boxes, 0 on a fourth).  So: bring the clocks up
// (~150 ms of stacks), time the free-running stack before AND after a ladder of beats on
// noise inputs, take the best beat + 0.1 us, and CONFIRM it against the free-running stack
// in two alternating rounds; it is kept only if it wins both by more than 1.5 %.  The
// answer is cached per (device, depth, FoV) for the process: ~0.3 s once.
struct PaceKey {
  int device, depth, V;
  bool operator<(const PaceKey& o) const {
    return device != o.device ? device < o.device : depth != o.depth ? depth < o.depth : V < o.V;
  }
};
struct PaceVal { int beat; float free_us, best_us; };
std::mutex g_pace_mu;
std::map<PaceKey, PaceVal> g_pace_cache;

int tune_pace(ffn_engine* e) {
  e->pace_auto = 0;
  e->pace_auto_us[0] = e->pace_auto_us[1] = 0.f;
  if (!(e->t_ok && e->flow_fits && e->flow == 2 && e->conv_variant == 9 && e->depth >= 2))
    return FFN_OK;
  const PaceKey key{e->device, e->depth, e->g.V};
  {
    std::lock_guard<std::mutex> lk(g_pace_mu);
    auto it = g_pace_cache.find(key);
    if (it != g_pace_cache.end()) {
      e->pace_auto = it->second.beat;
      e->pace_auto_us[0] = it->second.free_us;
      e->pace_auto_us[1] = it->second.best_us;
      return FFN_OK;
    }
  }
  const size_t V = (size_t)e->g.V;
  {
    std::vector<float> noise(2 * V);
    unsigned x = 12345u;
    for (auto& v : noise) {
      x = x * 1664525u + 1013904223u;
      v = ((int)(x >> 8) % 2001 - 1000) * 1e-3f;
    }
    HIP_TRY(hipMemcpy(e->up_image, noise.data(), V * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->up_seed, noise.data() + V, V * sizeof(float), hipMemcpyHostToDevice));
  }
  unsigned errs0 = 0;
  HIP_TRY(hipMemcpy(&errs0, e->flow_err, sizeof(errs0), hipMemcpyDeviceToHost));
  StepItems si;
  int rc = dense_items(e, 1, &si);
  if (rc) return rc;
  hipEvent_t ev0, ev1;
  HIP_TRY(hipEventCreate(&ev0));
  HIP_TRY(hipEventCreate(&ev1));
  const int user_pace = e->flow_pace;
  auto stacks_us = [&](int pace, int reps, float* us) -> int {
    e->flow_pace = pace;
    int r = FFN_OK;
    for (int k = 0; k < 6 && !r; ++k) r = run_stack(e, 1, si, std::nanf(""), INFINITY);
    if (!r && hipEventRecord(ev0, e->stream) != hipSuccess) r = FFN_ERR_HIP;
    for (int k = 0; k < reps && !r; ++k) r = run_stack(e, 1, si, std::nanf(""), INFINITY);
    if (!r && (hipEventRecord(ev1, e->stream) != hipSuccess ||
               hipEventSynchronize(ev1) != hipSuccess))
      r = FFN_ERR_HIP;
    float ms = 0.f;
    if (!r && hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) r = FFN_ERR_HIP;
    *us = ms * 1e3f / (float)reps;
    return r;
  };
  float us = 0.f, free_us = 0.f, best_us = 0.f;
  int best = 0;
  {  // clocks up: the board needs ~100 ms under load to settle
    const long long t_end = steady_ns() + 150000000LL;
    while (!rc && steady_ns() < t_end) rc = stacks_us(0, 60, &us);
  }
  if (!rc) rc = stacks_us(0, 40, &free_us);
  best_us = 1e30f;
  for (int pace = 560; pace <= 800 && !rc; pace += 20) {
    rc = stacks_us(pace, 24, &us);
    if (!rc && us < best_us) {
      best_us = us;
      best = pace;
    }
  }
  if (!rc) {
    rc = stacks_us(0, 40, &us);
    free_us = std::min(free_us, us);
  }
  bool keep = !rc && best > 0 && best_us < 0.985f * free_us;
  float conf_free = free_us, conf_paced = best_us;
  for (int round = 0; round < 2 && keep && !rc; ++round) {  // confirm, alternating
    float f = 0.f, p = 0.f;
    rc = stacks_us(0, 40, &f);
    if (!rc) rc = stacks_us(best + 10, 40, &p);
    keep = !rc && p < 0.985f * f;
    conf_free = f;
    conf_paced = p;
  }
  e->flow_pace = user_pace;
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  if (rc) return rc;
  unsigned errs = 0;
  HIP_TRY(hipMemcpy(&errs, e->flow_err, sizeof(errs), hipMemcpyDeviceToHost));
  const bool timed_out = errs != errs0;
  e->pace_auto_us[0] = conf_free;
  e->pace_auto_us[1] = keep ? conf_paced : best_us;
  if (!timed_out && keep) e->pace_auto = best + 10;
  if (!timed_out) {
    std::lock_guard<std::mutex> lk(g_pace_mu);
    g_pace_cache[key] = PaceVal{e->pace_auto, e->pace_auto_us[0], e->pace_auto_us[1]};
  }
  return FFN_OK;
}

int ffn_engine_set_weights(ffn_engine* e, const float* blob, size_t count) {
  EngineLock lock_(e);
  if (!e || !blob) return fail(FFN_ERR_ARG, "null argument");
  const size_t want = ffn_engine_weight_count(e->depth, kFeatures);
  if (count != want)
    return fail(FFN_ERR_ARG, "weight blob has %zu floats, expected %zu", count,
                want);
  drop_spec(e);
  HIP_TRY(hipSetDevice(e->device));
  const int F = kFeatures;
  std::vector<float> host(e->wl_off + F + 1 + 3, 0.0f);
  const float* src = blob;
  // conv0_a: [27][2][32] + bias, used as stored
  std::vector<uint16_t> hostd(e->wpackd_layer * (2 * e->depth - 1));  // zero tap 27
  std::vector<uint16_t> hosth(hostd.size());
  bool d_weights_ok = true;
  bool weights_in_fp16_range = true;
  // tap (kz', ky', kx') of the permuted layout = the original tap whose offset
  // along axis oa[a] is k'[a]
  int tap_of[27];
  for (int t = 0; t < 27; ++t) {
    const int kp[3] = {t / 9, (t / 3) % 3, t % 3};
    int ko[3];
    for (int a = 0; a < 3; ++a) ko[e->gp.oa[a]] = kp[a];
    tap_of[t] = ko[0] * 9 + ko[1] * 3 + ko[2];
  }
  std::memcpy(&host[e->w0a_off], src, sizeof(float) * 27 * 2 * F);
  for (int t = 0; t < 27; ++t)
    std::memcpy(&host[e->w0ap_off + (size_t)t * 2 * F], src + (size_t)tap_of[t] * 2 * F,
                sizeof(float) * 2 * F);
  src += 27 * 2 * F;
  std::memcpy(&host[e->b0a_off], src, sizeof(float) * F);
  src += F;
  // 32->32 convs: repack W[tap][ci][co] for the MFMA B operand:
  //   wpack[tap][nhalf][h][lane = 16*g + j][s] = W[tap][16h + 4g + s][16*nhalf + j]
  for (int l = 0; l < 2 * e->depth - 1; ++l) {
    float* wp = &host[e->wpack_off[l]];
    for (int tap = 0; tap < 27; ++tap)
      for (int nh = 0; nh < 2; ++nh)
        for (int h = 0; h < 2; ++h)
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 4; ++s) {
              const int gq = lane >> 4, j = lane & 15;
              const int ci = 16 * h + 4 * gq + s, co = 16 * nh + j;
              wp[((((size_t)tap * 2 + nh) * 2 + h) * 64 + lane) * 4 + s] =
                  src[((size_t)tap * F + ci) * F + co];
            }
    // conv32d / conv32m: W ~= hi + 2^-11 res (both fp16) as the 32x32x16 A
    // operand (rows = cout), + an all-zero tap 27
    //   wpackd[tap][khalf][plane hi, res][lane][c] =
    //       part(W[tap][16 khalf + 8 (lane >> 5) + c][lane & 31])
    {
      uint16_t* wd = &hostd[(size_t)l * e->wpackd_layer];
      for (int tap = 0; tap < 27; ++tap)
        for (int kh = 0; kh < 2; ++kh)
          for (int lane = 0; lane < 64; ++lane)
            for (int c = 0; c < 8; ++c) {
              const int ci = 16 * kh + 8 * (lane >> 5) + c, co = lane & 31;
              const float w = src[((size_t)tap_of[tap] * F + ci) * F + co];
              if (!(std::fabs(w) <= 65504.0f)) weights_in_fp16_range = false;
              uint16_t part[2];
              split_fp16x2(w, part);
              const size_t base = ((((size_t)tap * 2 + kh) * 2) * 64 + lane) * 8 + c;
              wd[base] = part[0];
              wd[base + 64 * 8] = part[1];
            }
    }
    // conv32h: the 16x16x32 A operand (rows = cout of half h, K = all 32 cin)
    //   wpackh[tap][h][plane hi, res][lane][c] = part(W[tap][8 (lane >> 4) + c][16 h + (lane & 15)])
    {
      uint16_t* wd = &hosth[(size_t)l * e->wpackd_layer];
      for (int tap = 0; tap < 27; ++tap)
        for (int h = 0; h < 2; ++h)
          for (int lane = 0; lane < 64; ++lane)
            for (int c = 0; c < 8; ++c) {
              const int ci = 8 * (lane >> 4) + c, co = 16 * h + (lane & 15);
              uint16_t part[2];
              split_fp16x2(src[((size_t)tap_of[tap] * F + ci) * F + co], part);
              const size_t base = ((((size_t)tap * 2 + h) * 2) * 64 + lane) * 8 + c;
              wd[base] = part[0];
              wd[base + 64 * 8] = part[1];
            }
    }
    src += 27 * F * F;
    std::memcpy(&host[e->bias_off[l]], src, sizeof(float) * F);
    src += F;
  }
  HIP_TRY(hipMemcpy(e->wpackd, hostd.data(), hostd.size() * sizeof(uint16_t),
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->wpackh, hosth.data(), hosth.size() * sizeof(uint16_t),
                    hipMemcpyHostToDevice));
  e->d_weights_ok = d_weights_ok;
  e->fp16_ok = weights_in_fp16_range;
  if ((!e->fp16_ok || !e->d_weights_ok) && e->conv_variant >= 6) {
    int rc = switch_variant(e, e->exact_variant);
    if (rc) return rc;
  }
  std::memcpy(&host[e->wl_off], src, sizeof(float) * (F + 1));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(e->weights, host.data(),
                    sizeof(float) * (e->wl_off + F + 1), hipMemcpyHostToDevice));
  e->weights_set = true;
  if (e->flow_pace < 0) {
    int rc = tune_pace(e);
    if (rc) return rc;
  }
  return FFN_OK;
}

int ffn_predict(ffn_engine* e, int n, const float* seed, const float* image,
                float* logits_out) {
  EngineLock lock_(e);
  if (!e || !seed || !image || !logits_out) return fail(FFN_ERR_ARG, "null argument");
  if (n < 1 || n > e->max_batch)
    return fail(FFN_ERR_ARG, "batch %d outside [1, %d]", n, e->max_batch);
  if (!e->weights_set) return fail(FFN_ERR_STATE, "weights not set");
  HIP_TRY(hipSetDevice(e->device));
  const size_t bytes = (size_t)n * e->g.V * sizeof(float);
  // through engine-owned pinned buffers: a pageable hipMemcpy costs ~80 us each
  float* h_seed = e->h_io;
  float* h_image = e->h_io + (size_t)e->max_batch * e->g.V;
  float* h_logits = e->h_io + 2 * (size_t)e->max_batch * e->g.V;
  std::memcpy(h_seed, seed, bytes);
  std::memcpy(h_image, image, bytes);
  HIP_TRY(hipMemcpyAsync(e->up_seed, h_seed, bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->up_image, h_image, bytes, hipMemcpyHostToDevice, e->stream));
  StepItems si;
  int rc = dense_items(e, n, &si);
  if (rc) return rc;
  // NaNs in a caller-provided seed stay NaN (the reference would feed them to TF).
  rc = run_stack(e, n, si, std::nanf(""), INFINITY);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h_logits, e->logits, bytes, hipMemcpyDeviceToHost,
                         e->stream));
  // A void run is repeated: after a time-out of the resident launch with per-layer
  // launches of the same kernels (same bits), after a range error -- also one that
  // only shows in that repeat -- with the exact-f32 kernel.  At most two repeats.
  for (int attempt = 0;; ++attempt) {
    unsigned flag[2] = {0, 0};
    if (e->conv_variant >= 6)
      HIP_TRY(hipMemcpyAsync(flag, e->range_flag, sizeof(flag), hipMemcpyDeviceToHost,
                             e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const bool is_void = e->conv_variant >= 6 && flag[0] == e->range_tag;
    if (!is_void) {
      if (e->last_stack_resident) e->flow_strikes = 0;
      break;
    }
    if (attempt == 2) return fail(FFN_ERR_HIP, "ffn_predict: the step stayed void");
    if (flow_voided(e, flag[1] == e->range_tag) == 0) {
      rc = switch_variant(e, e->exact_variant);
      if (rc) return rc;
    }
    rc = run_stack(e, n, si, std::nanf(""), INFINITY);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h_logits, e->logits, bytes, hipMemcpyDeviceToHost, e->stream));
  }
  std::memcpy(logits_out, h_logits, bytes);
  return FFN_OK;
}

int ffn_forward_resident(ffn_engine* e, int n, int repeats) {
  EngineLock lock_(e);
  if (!e) return fail(FFN_ERR_ARG, "null argument");
  if (n < 1 || n > e->max_batch)
    return fail(FFN_ERR_ARG, "batch %d outside [1, %d]", n, e->max_batch);
  if (!e->weights_set) return fail(FFN_ERR_STATE, "weights not set");
  HIP_TRY(hipSetDevice(e->device));
  StepItems si;
  int rc = dense_items(e, n, &si);
  if (rc) return rc;
  for (int r = 0; r < repeats; ++r) {
    rc = run_stack(e, n, si, std::nanf(""), INFINITY);
    if (rc) return rc;
  }
  return FFN_OK;
}

int ffn_engine_set_option(ffn_engine* e, const char* name, int value) {
  EngineLock lock_(e);
  if (!e || !name) return fail(FFN_ERR_ARG, "null argument");

  if (std::strcmp(name, "conv_variant") == 0) {
    if (value == -1) value = e->exact_variant;  // "the exact-f32 kernel of this FoV"
    if (value != 0 && value != 2 && !(value >= 6 && value <= 10))
      return fail(FFN_ERR_ARG, "conv_variant must be 0, 2, 6, 7, 8, 9 or 10 (1, 3, 4, 5 "
                               "were removed in ABI 7)");
    if (value == 10 && !e->h_ok)
      return fail(FFN_ERR_ARG, "conv_variant 10 unsupported for this fov / depth / device "
                               "(80-voxel workgroups, two per CU, all resident)");
    if (value >= 6 && e->weights_set && !e->fp16_ok)
      return fail(FFN_ERR_ARG, "conv_variant %d: a weight is outside the fp16 range",
                  value);
    if (value == 6 && !e->d_ok)
      return fail(FFN_ERR_ARG, "conv_variant 6 unsupported for this fov / depth");
    if (value == 7 && !e->e_ok)
      return fail(FFN_ERR_ARG, "conv_variant 7 unsupported for this fov / depth");
    if (value == 8 && !e->m_ok)
      return fail(FFN_ERR_ARG, "conv_variant 8 unsupported for this fov / depth");
    if (value == 9 && !e->t_ok)
      return fail(FFN_ERR_ARG, "conv_variant 9 unsupported for this fov / depth "
                               "(257 .. 512 chunks of 128 voxels)");
    if (value >= 6 && e->weights_set && !e->d_weights_ok)
      return fail(FFN_ERR_ARG, "conv_variant 6: a weight x 2^11 is outside the fp16 range");
    if (value == 2 && !(e->Rc == 256 || e->Rc == 288))
      return fail(FFN_ERR_ARG, "conv_variant %d unsupported for this fov", value);
    return switch_variant(e, value);
  }
  if (std::strcmp(name, "profile_every") == 0) {
    if (value < 1) return fail(FFN_ERR_ARG, "profile_every must be >= 1");
    e->prof_every = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "sync_mode") == 0) {
    if (value != 0 && value != 1) return fail(FFN_ERR_ARG, "sync_mode must be 0 or 1");
    e->sync_mode = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "debug_clock") == 0) {
    e->dbg_clock = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "batch_chunks") == 0) {
    // steps with >= 2 FoVs under conv_variant 6: 0 = the same kernel, 1 = its
    // 96-voxel form (bit-identical), 2 = conv32m (another summation order)
    if (value < 0 || value > 2) return fail(FFN_ERR_ARG, "batch_chunks 0..2");
    e->batch_chunks = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "tail_batched") == 0) {
    // conv_variant 9: 1 = steps with >= 2 FoVs also split off the tail (in
    // 96-voxel workgroups): every voxel then gets the same bits whatever the
    // batch, at 5-10 % of the batched rate
    e->tail_batched = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "flow_debug") == 0) {
    e->flow_debug = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "flow_pace") == 0) {
    if (value < -1 || value > 5000)
      return fail(FFN_ERR_ARG, "flow_pace: -1 (measured), 0 (off) .. 5000 (10-ns ticks)");
    e->flow_pace = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "flow_pace_spread") == 0) {
    if (value < -1 || value > 5000) return fail(FFN_ERR_ARG, "flow_pace_spread: -1 .. 5000");
    e->flow_pace_spread = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "flow_pace_tail") == 0) {
    if (value < 0 || value > 5000) return fail(FFN_ERR_ARG, "flow_pace_tail: 0 .. 5000");
    e->flow_pace_tail = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "flow") == 0) {
    // single-FoV steps of conv_variant 9: 0 one dependent launch per conv; 1 the
    // same launches with the flagged hand-off compiled in; 2 the resident stack
    // (conv32ps: one launch for all convs).  Same bits in every mode.
    if (value < 0 || value > 2) return fail(FFN_ERR_ARG, "flow: 0, 1 or 2");
    if (value && !e->t_ok)
      return fail(FFN_ERR_ARG, "flow needs conv32mt's geometry (conv_variant 9)");
    if (value && e->depth < 2) return fail(FFN_ERR_ARG, "flow needs depth >= 2");
    if (value == 2 && !e->flow_fits)
      return fail(FFN_ERR_ARG, "flow 2: this device cannot hold the resident launch's "
                  "workgroups all at once (compute units x occupancy)");
    drop_spec(e);
    e->flow = value;
    e->flow_auto_off = 0;
    e->flow_strikes = 0;
    e->flow_skip = 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "spec_force_mismatch") == 0) {
    e->spec_force_mismatch = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "fuse_paste") == 0) {
    e->fuse_paste = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "fuse_conv0a") == 0) {
    drop_spec(e);
    e->fuse_conv0a = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "paste_blocks") == 0) {
    if (value < 0 || value > 1024) return fail(FFN_ERR_ARG, "paste_blocks out of range");
    e->paste_blocks = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "debug_fused_twice") == 0) {
    e->debug_fused_twice = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "debug_submit_delay_ns") == 0) {
    e->debug_submit_delay_ns = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "stack_ahead") == 0) {
    // the next step's resident stack behind its speculative conv0_a, ahead of the host
    drop_spec(e);
    e->stack_ahead = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "speculate") == 0) {
    // single-FoV steps of ffn_canvas_segment_at: queue the next step's conv0_a
    // behind the paste, ahead of the host's turn-around (SpecArgs)
    drop_spec(e);
    e->speculate = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "stat_reset") == 0) {
    e->stat_spec_launched = e->stat_spec_hits = e->stat_spec_mismatch = 0;
    e->stat_ahead_used = e->stat_ahead_wasted = 0;
    e->stat_spec_miss_full = e->stat_spec_miss_short = 0;
    e->stat_segturn_queue_ns = e->stat_segturn_wait_ns = 0;
    e->stat_segturn_calls = 0;
    e->stat_calls = e->stat_items = 0;
    std::memset(e->stat_hist, 0, sizeof(e->stat_hist));
    e->stat_turn_host_ns = e->stat_launch_host_ns = 0;
    e->stat_turn_host_count = 0;
    e->t_arrived_ns = 0;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemset(e->d_stamps, 0, 32 * sizeof(long long)));
    return FFN_OK;
  }
  if (std::strcmp(name, "debug_fused_trace") == 0) {
    // the N-th single-FoV step from now stamps when its launches' roles ran
    // (ConvStackTab::stamps [4 .. 11]; read with debug_fused_stamp_K)
    // (minima at +inf, maxima at 0 -- here, not in the stream of the traced step)
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemset(e->d_stamps + 4, 0x7f, 4 * sizeof(long long)));
    HIP_TRY(hipMemset(e->d_stamps + 8, 0, 4 * sizeof(long long)));
    HIP_TRY(hipMemset(e->d_stamps + 12, 0, 20 * sizeof(long long)));
    HIP_TRY(hipMemset(e->d_stamps + 32, 0, 1024 * sizeof(long long)));
    e->fused_trace_in = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "debug_layer") == 0) {
    e->dbg_layer = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "ablate") == 0) {
    e->ablate = value;
    return FFN_OK;
  }
  if (std::strcmp(name, "fuse_head") == 0) {
    e->fuse_head = value != 0;
    return FFN_OK;
  }
  if (std::strcmp(name, "store_policy") == 0) {
    if (value < 0 || value > 2) return fail(FFN_ERR_ARG, "store_policy 0..2");
    e->store_policy = value;
    return FFN_OK;
  }
  return fail(FFN_ERR_ARG, "unknown option '%s'", name);
}

int ffn_engine_get_option(ffn_engine* e, const char* name, int* value) {
  EngineLock lock_(e);
  if (!e || !name || !value) return fail(FFN_ERR_ARG, "null argument");
  if (std::strcmp(name, "conv_variant") == 0) *value = e->conv_variant;
  else if (std::strcmp(name, "fuse_head") == 0) *value = e->fuse_head;
  else if (std::strcmp(name, "exact_variant") == 0) *value = e->exact_variant;
  else if (std::strcmp(name, "stat_step_calls") == 0) *value = (int)e->stat_calls;
  else if (std::strcmp(name, "stat_step_items") == 0) *value = (int)e->stat_items;
  else if (std::strcmp(name, "speculate") == 0) *value = e->speculate;
  else if (std::strcmp(name, "stack_ahead") == 0) *value = e->stack_ahead;
  else if (std::strcmp(name, "paste_blocks") == 0) *value = e->paste_blocks;
  else if (std::strcmp(name, "stat_segturn_queue_ns") == 0)
    *value = e->stat_segturn_calls ? (int)(e->stat_segturn_queue_ns / e->stat_segturn_calls) : 0;
  else if (std::strcmp(name, "stat_segturn_wait_ns") == 0)
    *value = e->stat_segturn_calls ? (int)(e->stat_segturn_wait_ns / e->stat_segturn_calls) : 0;
  else if (std::strcmp(name, "stat_segturn_calls") == 0) *value = (int)e->stat_segturn_calls;
  else if (std::strcmp(name, "stat_spec_miss_full") == 0) *value = (int)e->stat_spec_miss_full;
  else if (std::strcmp(name, "stat_spec_miss_short") == 0) *value = (int)e->stat_spec_miss_short;
  else if (std::strcmp(name, "stat_ahead_used") == 0) *value = (int)e->stat_ahead_used;
  else if (std::strcmp(name, "stat_ahead_wasted") == 0) *value = (int)e->stat_ahead_wasted;
  else if (std::strcmp(name, "fuse_paste") == 0) *value = e->fuse_paste;
  else if (std::strcmp(name, "fuse_conv0a") == 0) *value = e->fuse_conv0a;
  else if (std::strcmp(name, "stat_spec_launched") == 0)
    *value = (int)e->stat_spec_launched;
  else if (std::strcmp(name, "stat_spec_hits") == 0) *value = (int)e->stat_spec_hits;
  else if (std::strcmp(name, "stat_spec_mismatch") == 0)
    *value = (int)e->stat_spec_mismatch;
  else if (std::strcmp(name, "stat_many_carried") == 0)
    *value = (int)e->stat_many_carried;
  else if (std::strcmp(name, "flow") == 0) *value = e->flow;
  else if (std::strcmp(name, "flow_auto_off") == 0) *value = e->flow_auto_off;
  else if (std::strcmp(name, "stat_flow_voids") == 0) *value = (int)e->stat_flow_voids;
  else if (std::strcmp(name, "stat_turn_host_ns") == 0)
    *value = e->stat_turn_host_count
                 ? (int)(e->stat_turn_host_ns / e->stat_turn_host_count) : 0;
  else if (std::strcmp(name, "stat_launch_host_ns") == 0)
    *value = e->stat_turn_host_count
                 ? (int)(e->stat_launch_host_ns / e->stat_turn_host_count) : 0;
  else if (std::strcmp(name, "stat_turn_count") == 0) *value = (int)e->stat_turn_host_count;
  else if (std::strcmp(name, "stat_ahead_aborted") == 0) {
    long long st[4] = {0, 0, 0, 0};
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(st, e->d_stamps, sizeof(st), hipMemcpyDeviceToHost));
    *value = (int)st[3];
  }
  else if (std::strcmp(name, "stat_turn_gpu_ns") == 0) {
    long long st[4] = {0, 0, 0, 0};
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(st, e->d_stamps, sizeof(st), hipMemcpyDeviceToHost));
    *value = st[2] ? (int)(st[1] * 10 / st[2]) : 0;
  }
  else if (std::strncmp(name, "debug_fused_stamp_", 18) == 0) {
    // stamp k (4 .. 11) minus the stack's first entry [7], in 10-ns ticks
    const int k = std::atoi(name + 18);
    long long st[32 + 1024];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(st, e->d_stamps, sizeof(st), hipMemcpyDeviceToHost));
    for (int w = 0; w < 512; ++w) st[25] = std::max(st[25], st[32 + w]);  // (per-workgroup ends)
    {  // [27]: main workgroups that share their CU with another main workgroup
      std::map<unsigned, int> per_cu;
      for (int w = 0; w < 512; ++w) {
        const unsigned v = (unsigned)st[32 + 512 + w];
        if (st[32 + 512 + w] != 0 && ((v - 1) >> 31)) per_cu[(v - 1) & 0x7fffffffu] += 1;
      }
      long long shared = 0;
      for (const auto& kv : per_cu) if (kv.second > 1) shared += kv.second;
      st[27] = st[4] + shared;  // (the getter subtracts the origin)
      if (!e->stack_ahead) st[27] = st[7] + shared;
    }
    if (k < 4 || k > 27 || k == 15) return fail(FFN_ERR_ARG, "debug_fused_stamp_4 .. 27");
    // (stack_ahead: the stack inside the traced window is the NEXT step's; the fused
    // launch's first faces entry [4] is the origin then)
    *value = (int)(st[k] - (e->stack_ahead ? st[4] : st[7]));
  }
  else if (std::strcmp(name, "flow_pace") == 0) *value = e->flow_pace;
  else if (std::strcmp(name, "flow_pace_now") == 0)
    *value = e->flow_pace < 0 ? e->pace_auto : e->flow_pace;
  else if (std::strcmp(name, "flow_pace_free_ns") == 0) *value = (int)(e->pace_auto_us[0] * 1e3f);
  else if (std::strcmp(name, "flow_pace_best_ns") == 0) *value = (int)(e->pace_auto_us[1] * 1e3f);
  else if (std::strcmp(name, "stat_flow_timeouts") == 0) {
    unsigned v = 0;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(&v, e->flow_err, sizeof(v), hipMemcpyDeviceToHost));
    *value = (int)v;
  }
  else if (std::strncmp(name, "stat_hist_", 10) == 0) {
    const int k = std::atoi(name + 10);
    if (k < 0 || k > 64) return fail(FFN_ERR_ARG, "stat_hist_<0..64>");
    *value = (int)e->stat_hist[k];
  }
  else if (std::strcmp(name, "store_policy") == 0) *value = e->store_policy;
  else if (std::strcmp(name, "sync_mode") == 0) *value = e->sync_mode;
  else if (std::strcmp(name, "profile_every") == 0) *value = e->prof_every;
  else return fail(FFN_ERR_ARG, "unknown option '%s'", name);
  return FFN_OK;
}

int ffn_engine_set_pred_size(ffn_engine* e, const int32_t pred_zyx[3]) {
  EngineLock lock_(e);
  if (!e || !pred_zyx) return fail(FFN_ERR_ARG, "null argument");
  Geom& g = e->g;
  const int f[3] = {g.fz, g.fy, g.fx}, d[3] = {g.dz, g.dy, g.dx};
  for (int a = 0; a < 3; ++a) {
    if (pred_zyx[a] < 1 || pred_zyx[a] > f[a])
      return fail(FFN_ERR_ARG, "pred size %d outside 1..%d (axis %d)", pred_zyx[a],
                  f[a], a);
    if (d[a] > pred_zyx[a] / 2)
      return fail(FFN_ERR_ARG, "delta %d beyond the prediction's half size %d (axis %d)",
                  d[a], pred_zyx[a] / 2, a);
    // the reference pastes into [start + delta, end - delta) with delta = (seed -
    // pred) // 2 (inference.py:218,410-411): an odd difference gives it a box of
    // pred + 1 voxels and a shape mismatch -- not a geometry it can run
    if ((f[a] - pred_zyx[a]) % 2 != 0)
      return fail(FFN_ERR_ARG, "seed size %d - pred size %d is odd (axis %d): the "
                  "prediction cannot be centred in the seed FoV", f[a], pred_zyx[a], a);
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  drop_spec(e);
  g.Vp = 1;
  g.crop = 0;
  for (int a = 0; a < 3; ++a) {
    // update_seed's zero padding (model.py:168-183): (seed - pred) // 2 in front
    g.c0[a] = (f[a] - pred_zyx[a]) / 2;
    g.c1[a] = g.c0[a] + pred_zyx[a];
    g.Vp *= pred_zyx[a];
    g.crop |= pred_zyx[a] != f[a];
  }
  return FFN_OK;
}

int ffn_engine_debug_flow_trace(ffn_engine* e, long long* out, int max_slots) {
  EngineLock lock_(e);
  if (!e || !out) return fail(FFN_ERR_ARG, "null argument");
  if (!e->flow_trace) return fail(FFN_ERR_ARG, "no flow trace on this engine (conv_variant 9)");
  if (max_slots < 0 || max_slots > e->flow_trace_slots)
    return fail(FFN_ERR_ARG, "max_slots must be 0..%d", e->flow_trace_slots);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const size_t row = (size_t)kFlowTraceLayers * 8 * sizeof(long long);
  HIP_TRY(hipMemcpy(out, e->flow_trace, row * max_slots, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(e->flow_trace, 0, row * e->flow_trace_slots));
  return FFN_OK;
}

int ffn_engine_debug_workgroups(ffn_engine* e, long long* out, int max_wgs) {
  EngineLock lock_(e);
  if (!e || !out) return fail(FFN_ERR_ARG, "null argument");
  if (max_wgs < 0 || max_wgs > kDbgMaxWgs)
    return fail(FFN_ERR_ARG, "max_wgs must be 0..%d", kDbgMaxWgs);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(out, e->d_dbg + 24, (size_t)4 * max_wgs * sizeof(long long),
                    hipMemcpyDeviceToHost));
  // the next run starts from zeros (a workgroup that did not run leaves them)
  HIP_TRY(hipMemset(e->d_dbg + 24, 0, (size_t)4 * kDbgMaxWgs * sizeof(long long)));
  return FFN_OK;
}

int ffn_engine_debug_clocks(ffn_engine* e, long long* out24) {
  EngineLock lock_(e);
  if (!e || !out24) return fail(FFN_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(out24, e->d_dbg, 24 * sizeof(long long), hipMemcpyDeviceToHost));
  return FFN_OK;
}

int ffn_engine_synchronize(ffn_engine* e) {
  EngineLock lock_(e);
  if (!e) return fail(FFN_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return FFN_OK;
}

int ffn_engine_set_profiling(ffn_engine* e, int mode) {
  EngineLock lock_(e);
  if (!e) return fail(FFN_ERR_ARG, "null argument");
  if (mode < 0 || mode > 2) return fail(FFN_ERR_ARG, "mode must be 0, 1 or 2");
  HIP_TRY(hipSetDevice(e->device));
  int rc = flush_events(e);
  if (rc) return rc;
  e->chain_launches_pending.clear();
  e->prof_mode = mode;
  return FFN_OK;
}

int ffn_engine_get_profile(ffn_engine* e, double* conv_ms_total,
                           int64_t* conv_launches, int reset) {
  EngineLock lock_(e);
  if (!e) return fail(FFN_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->device));
  int rc = flush_events(e);
  if (rc) return rc;
  if (conv_ms_total) *conv_ms_total = e->conv_ms;
  if (conv_launches) *conv_launches = e->conv_launches;
  if (reset) {
    e->conv_ms = 0.0;
    e->conv_launches = 0;
    e->prof_samples.clear();
  }
  return FFN_OK;
}

int ffn_engine_get_profile_samples(ffn_engine* e, float* out_ms, int max_n, int* n) {
  EngineLock lock_(e);
  if (!e || !n || (max_n > 0 && !out_ms)) return fail(FFN_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->device));
  int rc = flush_events(e);
  if (rc) return rc;
  const int k = (int)std::min<size_t>(e->prof_samples.size(), (size_t)std::max(max_n, 0));
  for (int i = 0; i < k; ++i) out_ms[i] = e->prof_samples[i];
  *n = (int)e->prof_samples.size();
  return FFN_OK;
}

/* ------------------------------- canvas ---------------------------------- */

namespace {

// image_f32 (already normalised) or image_u8 + the normalisation constants
int canvas_create(ffn_engine* e, const float* image_f32, const uint8_t* image_u8,
                  float mean, float stddev, const int32_t shape_zyx[3],
                  ffn_canvas** out) {
  EngineLock lock_(e);
  if (!e || (!image_f32 && !image_u8) || !shape_zyx || !out)
    return fail(FFN_ERR_ARG, "null argument");
  *out = nullptr;
  for (int k = 0; k < 3; ++k)
    if (shape_zyx[k] < 1) return fail(FFN_ERR_ARG, "bad canvas shape");
  if (image_u8 && !(stddev != 0.0f))
    return fail(FFN_ERR_ARG, "image_stddev must be non-zero");
  HIP_TRY(hipSetDevice(e->device));
  ffn_canvas* c = new ffn_canvas();
  c->engine = e;
  c->cz = shape_zyx[0];
  c->cy = shape_zyx[1];
  c->cx = shape_zyx[2];
  c->nvox = (size_t)c->cz * c->cy * c->cx;
  hipError_t err = hipSuccess;
  if (image_f32) {
    err = hipMalloc(&c->image, c->nvox * sizeof(float));
    if (err == hipSuccess)
      err = hipMemcpy(c->image, image_f32, c->nvox * sizeof(float),
                      hipMemcpyHostToDevice);
  } else {
    // (image.astype(np.float32) - image_mean) / image_stddev (runner.py:383-385):
    // two correctly rounded f32 operations per value, evaluated here for the
    // 256 possible inputs -- the device only looks the result up, so a uint8
    // canvas is bit-identical to the f32 one by construction
    float lut[256];
    for (int v = 0; v < 256; ++v) {
      volatile float d = (float)v - mean;  // volatile: no fused / widened form
      lut[v] = d / stddev;
    }
    err = hipMalloc(&c->image_u8, c->nvox);
    if (err == hipSuccess) err = hipMalloc(&c->image_lut, sizeof(lut));
    if (err == hipSuccess)
      err = hipMemcpy(c->image_u8, image_u8, c->nvox, hipMemcpyHostToDevice);
    if (err == hipSuccess)
      err = hipMemcpy(c->image_lut, lut, sizeof(lut), hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) err = hipMalloc(&c->seed, c->nvox * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(&c->seg, c->nvox * sizeof(int32_t));
  if (err == hipSuccess) err = hipMemset(c->seg, 0, c->nvox * sizeof(int32_t));
  if (err == hipSuccess) {
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, e->stream,
                       reinterpret_cast<uint32_t*>(c->seed), 0x7fc00000u, c->nvox);
    err = hipStreamSynchronize(e->stream);
  }
  if (err != hipSuccess) {
    int rc = fail(FFN_ERR_HIP, "canvas allocation failed: %s", hipGetErrorString(err));
    ffn_canvas_destroy(c);
    return rc;
  }
  {
    const hipError_t ee = hipEventCreateWithFlags(&c->ev_util, hipEventDisableTiming);
    if (ee != hipSuccess) {
      int rc = fail(FFN_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(ee));
      ffn_canvas_destroy(c);
      return rc;
    }
  }
  e->canvases.push_back(c);
  *out = c;
  return FFN_OK;
}

}  // namespace

int ffn_canvas_create(ffn_engine* e, const float* image_f32,
                      const int32_t shape_zyx[3], ffn_canvas** out) {
  if (!image_f32) return fail(FFN_ERR_ARG, "null argument");
  return canvas_create(e, image_f32, nullptr, 0.f, 1.f, shape_zyx, out);
}

int ffn_canvas_create_u8(ffn_engine* e, const uint8_t* image_u8,
                         const int32_t shape_zyx[3], float image_mean,
                         float image_stddev, ffn_canvas** out) {
  if (!image_u8) return fail(FFN_ERR_ARG, "null argument");
  return canvas_create(e, nullptr, image_u8, image_mean, image_stddev, shape_zyx,
                       out);
}

namespace {
int resolve_many_carry(ffn_engine* e, const ffn_canvas* only = nullptr);
}

void ffn_canvas_destroy(ffn_canvas* c) {
  if (c && c->engine) (void)resolve_many_carry(c->engine, c);  // a step in flight
  EngineLock lock_(c ? c->engine : nullptr);
  if (!c) return;
  if (c->engine) {
    ffn_engine* e = c->engine;
    drop_spec(e);
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamSynchronize(e->ustream);
    if (c->ev_util) (void)hipEventDestroy(c->ev_util);
    e->canvases.erase(std::remove(e->canvases.begin(), e->canvases.end(), c),
                      e->canvases.end());
    (void)hipFree(c->image);
    (void)hipFree(c->image_u8);
    (void)hipFree(c->image_lut);
    (void)hipFree(c->seed);
    (void)hipFree(c->seg);
  }
  delete c;
}

int ffn_canvas_init_seed(ffn_canvas* c, const int32_t pos[3], float value) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !pos) return fail(FFN_ERR_ARG, "null argument");
  if (pos[0] < 0 || pos[0] >= c->cz || pos[1] < 0 || pos[1] >= c->cy ||
      pos[2] < 0 || pos[2] >= c->cx)
    return fail(FFN_ERR_ARG, "seed position outside the canvas");
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c));
  if (c->dirty_lo[0] < c->dirty_hi[0]) {
    long total = 0;
    Box b = make_box(c, c->dirty_lo, c->dirty_hi, &total);
    if ((size_t)total * 2 >= c->nvox) {  // most of the volume: linear fill
      hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, e->ustream,
                         reinterpret_cast<uint32_t*>(c->seed), 0x7fc00000u,
                         c->nvox);
    } else if (total > 0) {
      hipLaunchKernelGGL((box_fill_kernel<uint32_t>), dim3(grid_for(total)),
                         dim3(256), 0, e->ustream,
                         reinterpret_cast<uint32_t*>(c->seed), b, total,
                         0x7fc00000u);
    }
  }
  {
    const int lo[3] = {pos[0], pos[1], pos[2]};
    const int hi[3] = {pos[0] + 1, pos[1] + 1, pos[2] + 1};
    c->dirty_lo[0] = c->dirty_hi[0] = 0;  // clean, then the seed point itself
    c->mark_dirty(lo, hi);
  }
  const size_t ci = ((size_t)pos[0] * c->cy + pos[1]) * c->cx + pos[2];
  hipLaunchKernelGGL(set_seed_point_kernel, dim3(1), dim3(1), 0, e->ustream,
                     c->seed, ci, value);
  HIP_TRY(hipGetLastError());
  HIP_TRY(lock_.end(e, c));
  return FFN_OK;
}

int ffn_canvas_step_submit(ffn_engine* e, int n, ffn_canvas* const* canvases,
                           const ffn_step_request* requests,
                           const ffn_step_params* params, uint32_t* ticket) {
  EngineLock lock_(e);
  if (!e || !canvases || !requests || !params || !ticket)
    return fail(FFN_ERR_ARG, "null argument");
  if (n < 1 || n > e->max_batch)
    return fail(FFN_ERR_ARG, "batch %d outside [1, %d]", n, e->max_batch);
  if (!e->weights_set) return fail(FFN_ERR_STATE, "weights not set");
  // the slot after the last one used, else the other one (two host threads
  // need not alternate)
  const int slot = e->slot_n[e->next_slot] == 0 ? e->next_slot : e->next_slot ^ 1;
  if (e->slot_n[slot] != 0)
    return fail(FFN_ERR_STATE,
                "two steps already in flight: call ffn_canvas_step_wait first");
  HIP_TRY(hipSetDevice(e->device));
  const Geom& g = e->g;
  // the segment loop's hint belongs to THIS call (a single FoV) or to none
  int hint_n = 0;
  int hint_pos[kSpecMax][3];
  bool from_loop = false;
  for (int k = 0; k < n; ++k)
    if (canvases[k]) {
      if (n == 1) {
        hint_n = canvases[k]->hint_n;
        std::memcpy(hint_pos, canvases[k]->hint_pos, sizeof(hint_pos));
        from_loop = canvases[k]->hint_from_loop;
      }
      canvases[k]->hint_n = 0;
      canvases[k]->hint_from_loop = false;
    }
  StepItem* h_items = e->h_items + (size_t)slot * e->max_batch;
  StepItem* d_items = e->d_items + (size_t)slot * e->max_batch;
  unsigned long long* h_pub = e->h_pub + (size_t)slot * e->max_batch * kPubWords;
  const int other = slot ^ 1;
  for (int k = 0; k < n; ++k) {
    const ffn_canvas* c = canvases[k];
    if (!c || c->engine != e) return fail(FFN_ERR_ARG, "canvas %d not of this engine", k);
    const ffn_step_request& r = requests[k];
    const int half[3] = {g.fz / 2, g.fy / 2, g.fx / 2};
    const int dims[3] = {c->cz, c->cy, c->cx};
    for (int a = 0; a < 3; ++a)
      if (r.pos[a] - half[a] < 0 || r.pos[a] + half[a] >= dims[a])
        return fail(FFN_ERR_ARG, "FoV at (%d,%d,%d) leaves the canvas", r.pos[0],
                    r.pos[1], r.pos[2]);
    if (r.num_candidates < 0 || r.num_candidates > FFN_MAX_CANDIDATES)
      return fail(FFN_ERR_ARG, "num_candidates out of range");
    for (int k2 = 0; k2 < k; ++k2)
      if (canvases[k2] == c)
        return fail(FFN_ERR_ARG, "canvas appears twice in one batch");
    // a canvas steps sequentially: it cannot also be in the step in flight
    for (int k2 = 0; k2 < e->slot_n[other]; ++k2)
      if (e->slot_canvas[other][k2] == c)
        return fail(FFN_ERR_STATE, "canvas %d already has a step in flight", k);
    {
      const int lo[3] = {r.pos[0] - half[0], r.pos[1] - half[1], r.pos[2] - half[2]};
      const int hi[3] = {r.pos[0] + half[0] + 1, r.pos[1] + half[1] + 1,
                         r.pos[2] + half[2] + 1};
      const_cast<ffn_canvas*>(c)->mark_dirty(lo, hi);
    }
    if (c->util_pending) {  // its last commit / re-seed, on the utility stream
      HIP_TRY(hipStreamWaitEvent(e->stream, c->ev_util, 0));
      const_cast<ffn_canvas*>(c)->util_pending = false;
    }
    // a single-FoV step raises its flag before it pastes (below)
    const_cast<ffn_canvas*>(c)->paste_after_flag = n == 1;
    StepItem& it = h_items[k];
    it.image = c->image;
    it.image_u8 = c->image_u8;
    it.image_lut = c->image_lut;
    it.seed = c->seed;
    it.seg = c->seg;
    it.cz = c->cz;
    it.cy = c->cy;
    it.cx = c->cx;
    it.req = r;
  }
  HIP_TRY(hipSetDevice(e->device));
  StepItems si;
  si.items = d_items;
  si.use_inline = n == 1;
  si.inline_item = h_items[0];
  if (n > 1)
    HIP_TRY(hipMemcpyAsync(d_items, h_items, sizeof(StepItem) * n,
                           hipMemcpyHostToDevice, e->stream));
  // the speculative conv0_a queued behind the last step, if this is the step it
  // was made for (same canvas and parameters, a position on its list): the
  // device took the first valid position of that list, and so did the caller
  // (only the loop's own steps: it pops the FIRST valid position of its list; a
  // caller stepping the canvas itself may pick any)
  int spec_expected = -1;
  if (e->spec.valid && n == 1 && from_loop && e->spec.canvas == canvases[0] &&
      e->spec.variant == e->conv_variant &&
      std::memcmp(&e->spec.pad_value, &params->pad_value, sizeof(float)) == 0 &&
      std::memcmp(&e->spec.move_thr, &params->move_threshold, sizeof(float)) == 0)
    for (int j = 0; j < e->spec.n && spec_expected < 0; ++j)
      if (std::memcmp(e->spec.pos[j], requests[0].pos, sizeof(int) * 3) == 0)
        spec_expected = j;
  if (spec_expected >= 0) e->stat_spec_hits += 1;
  else if (e->spec.valid && n == 1 && from_loop && e->spec.canvas == canvases[0]) {
    // a launch made ahead that this step does not run on: was its hint list full (the
    // queue held more candidates than a launch looks at) or short (the step comes from the
    // moves the last step queued)?
    if (e->spec.n >= kSpecMax) e->stat_spec_miss_full += 1;
    else e->stat_spec_miss_short += 1;
  }
  if (spec_expected >= 0 && e->spec_force_mismatch > 0) {  // test hook
    e->spec_force_mismatch -= 1;
    spec_expected = kSpecMax;  // an index the device cannot have chosen
  }
  const bool traced = n == 1 && e->fused_trace_in > 0 && --e->fused_trace_in == 0;
  e->trace_now = traced;  // (the launches of this call stamp: a kernel argument)
  if (e->debug_submit_delay_ns > 0) {
    const long long t_d = steady_ns();
    while (steady_ns() - t_d < e->debug_submit_delay_ns) {}
  }
  int rc = FFN_OK;
  if (e->ahead_valid && spec_expected >= 0) {
    // this step's stack is in the stream already, behind the conv0_a made for it
    // (what run_stack would have done was done when it was queued)
    e->ahead_valid = false;
    e->spec.valid = false;
    e->stat_ahead_used += 1;
    if (e->ahead_ev_pending) {  // its event pair is a stack's: among the samples from now on
      e->ahead_ev_pending = false;
      if (e->events_used + 2 > (int)e->events.size()) {
        rc = flush_events(e);
        if (rc) return rc;
      }
      e->chain_launches_pending.push_back(2 * e->depth - 1);
      std::swap(e->events[e->events_used++], e->ahead_ev[0]);
      std::swap(e->events[e->events_used++], e->ahead_ev[1]);
    }
  } else {
    rc = run_stack(e, n, si, params->pad_value, params->move_threshold,
                   spec_expected >= 0);
  }
  if (rc) return rc;
  const bool this_stack_resident = e->last_stack_resident;
  const unsigned step_id = ++e->step_id ? e->step_id : ++e->step_id;  // never 0
  // One FoV: faces first -- the host's turn-around is on the critical path and
  // the paste runs under it.  Several: paste first, so that the completion flag
  // the host waits for also says "pasted" and a canvas whose segment has ended
  // can be committed on the utility stream at once (ffn_engine::ustream).
  auto paste = [&]() {
    hipLaunchKernelGGL(paste_kernel, dim3(71, n), dim3(512), 0, e->stream, si, g,
                       e->logits, e->seed_raw, e->count, e->count_blocks,
                       params->move_threshold, params->disco_seed_threshold,
                       e->range_flag, e->range_tag,
                       e->d_spec_choice, spec_expected);
  };
  // the next step's conv0_a, behind the paste and ahead of the host's turn-around
  // (SpecArgs): for the positions the segment loop expects to pop next
  SpecArgs sp;
  sp.n = 0;
  sp.move_thr = params->move_threshold;
  sp.choice = nullptr;  // (launch_conv0a / conv0a_split_args: the next step's word)
  if (n == 1) {
    const ffn_canvas* c = canvases[0];
    const int half[3] = {g.fz / 2, g.fy / 2, g.fx / 2};
    const int dims[3] = {c->cz, c->cy, c->cx};
    for (int j = 0; j < hint_n && e->speculate; ++j) {
      bool inside = true;
      for (int a = 0; a < 3; ++a)
        if (hint_pos[j][a] - half[a] < 0 || hint_pos[j][a] + half[a] >= dims[a])
          inside = false;
      if (!inside) continue;
      for (int a = 0; a < 3; ++a) sp.pos[sp.n][a] = hint_pos[j][a];
      ++sp.n;
    }
    for (int j = sp.n; j < kSpecMax && sp.n > 0; ++j)  // unused slots: readable positions
      for (int a = 0; a < 3; ++a) sp.pos[j][a] = sp.pos[0][a];
  }
  // ... in the SAME launch as this step's faces and paste where it can be
  // (faces_paste_conv0a_kernel: it gathers the canvas as the paste is leaving it)
  const bool fused_next = n == 1 && sp.n > 0 && e->fuse_paste && e->fuse_conv0a &&
                          e->conv_variant >= 6;
  if (n > 1) paste();
  if (fused_next) {
    Conv0Next nx;
    const int tiles = conv0a_split_args(e, params->pad_value, next_tag(e->range_tag), sp,
                                        true, nx);
    const int paste_blocks = e->paste_blocks > 0
        ? e->paste_blocks
        : std::max(kPasteBlocksMin, std::min(kPasteBlocks, e->cus - 1 - tiles));
    // (experiment debug_fused_twice: the same launch made twice -- idempotent: same record,
    // same paste, same conv0_a -- so that the second, traced one starts on warm caches)
    if (e->debug_fused_twice)
      hipLaunchKernelGGL(faces_paste_conv0a_kernel, dim3(1 + paste_blocks + tiles),
                         dim3(512), 0, e->stream, si, g, e->logits, e->seed_raw, e->count,
                         e->count_blocks, params->move_threshold,
                         params->disco_seed_threshold, params->deleted_threshold,
                         e->range_flag, e->range_tag, h_pub, step_id, e->d_spec_choice,
                         spec_expected, nx, e->d_stamps, 0, paste_blocks);
    hipLaunchKernelGGL(faces_paste_conv0a_kernel, dim3(1 + paste_blocks + tiles),
                       dim3(512), 0, e->stream, si, g, e->logits, e->seed_raw, e->count,
                       e->count_blocks, params->move_threshold,
                       params->disco_seed_threshold, params->deleted_threshold,
                       e->range_flag, e->range_tag, h_pub, step_id, e->d_spec_choice,
                       spec_expected, nx, e->d_stamps, e->trace_now ? 1 : 0, paste_blocks);
  } else if (n == 1 && e->fuse_paste) {
    hipLaunchKernelGGL(faces_paste_kernel, dim3(1 + 71), dim3(512), 0, e->stream, si,
                       g, e->logits, e->seed_raw, e->count, e->count_blocks,
                       params->move_threshold, params->disco_seed_threshold,
                       params->deleted_threshold, e->range_flag, e->range_tag,
                       h_pub, step_id, e->d_spec_choice, spec_expected);
  } else {
    hipLaunchKernelGGL(faces_kernel, dim3(n), dim3(512), 0, e->stream, si, g,
                       e->logits, e->seed_raw, e->count, e->count_blocks,
                       params->move_threshold, params->disco_seed_threshold,
                       params->deleted_threshold, e->range_flag, e->range_tag,
                       h_pub, step_id, e->d_spec_choice, spec_expected);
    if (n == 1) paste();
  }
  if (n == 1) {
    ffn_canvas* c = canvases[0];
    if (sp.n > 0) {
      if (!fused_next)
        launch_conv0a(e, 1, si, params->pad_value, next_tag(e->range_tag), sp, true);
      // ... and that step's stack behind it (stack_ahead): only where this step's own
      // stack was a resident launch that is still trusted, and the loop made the step
      bool ahead = false;
      if (fused_next && e->stack_ahead && from_loop && this_stack_resident &&
          e->flow == 2 && e->flow_skip == 0 && e->conv_variant == 9) {
        rc = run_stack(e, 1, si, params->pad_value, params->move_threshold, true, true);
        if (rc) return rc;
        ahead = true;
      }
      e->ahead_valid = ahead;
      e->spec.valid = true;
      e->spec.canvas = c;
      e->spec.n = sp.n;
      std::memcpy(e->spec.pos, sp.pos, sizeof(sp.pos));
      e->spec.pad_value = params->pad_value;
      e->spec.move_thr = params->move_threshold;
      e->spec.variant = e->conv_variant;
      e->stat_spec_launched += 1;
    }
  }
  e->trace_now = false;
  HIP_TRY(hipGetLastError());
  e->stat_calls += 1;
  e->stat_items += n;
  e->stat_hist[n < 64 ? n : 64] += 1;
  e->slot_n[slot] = n;
  e->slot_resident[slot] = this_stack_resident;
  e->slot_ticket[slot] = step_id;
  e->slot_canvas[slot].assign(canvases, canvases + n);
  e->next_slot = other;
  *ticket = step_id;
  return FFN_OK;
}

namespace {
int step_wait_impl(ffn_engine* e, uint32_t ticket, ffn_step_result* results,
                   bool* spec_mismatch);
}

int ffn_canvas_step_wait(ffn_engine* e, uint32_t ticket,
                         ffn_step_result* results) {
  return step_wait_impl(e, ticket, results, nullptr);
}

namespace {
// spec_mismatch: where to report a step that ran on a speculative conv0_a made
// for another position (it pasted nothing; ffn_canvas_step repeats it) -- NULL:
// such a step is an error
int step_wait_impl(ffn_engine* e, uint32_t ticket, ffn_step_result* results,
                   bool* spec_mismatch) {
  if (spec_mismatch) *spec_mismatch = false;
  if (!e || !results) return fail(FFN_ERR_ARG, "null argument");
  int slot = -1;
  int n = 0;
  {
    EngineLock lock_(e);
    for (int s = 0; s < 2; ++s)
      if (e->slot_n[s] != 0 && e->slot_ticket[s] == ticket && !e->slot_waited[s])
        slot = s;
    if (slot < 0)
      return fail(FFN_ERR_STATE, "no step with ticket %u in flight", ticket);
    n = e->slot_n[slot];
    e->slot_waited[slot] = true;
  }
  const unsigned step_id = ticket;
  unsigned long long* h_pub = e->h_pub + (size_t)slot * e->max_batch * kPubWords;
  // The slot -- its descriptor, result and flag arrays -- stays taken until the
  // results have been copied out (another thread may submit meanwhile); it is
  // free again whatever happens below.  No lock while waiting: that wait is
  // what two host threads overlap.
  struct SlotRelease {
    ffn_engine* e;
    int slot;
    ~SlotRelease() {
      EngineLock lock_(e);
      e->slot_n[slot] = 0;
      e->slot_waited[slot] = false;
      e->slot_canvas[slot].clear();
    }
  } release_{e, slot};
  HIP_TRY(hipSetDevice(e->device));
  // every published word of the n records carries this step's number
  auto arrived = [&]() {
    for (int k = 0; k < n * kPubWords; ++k)
      if ((unsigned)(__atomic_load_n(&h_pub[k], __ATOMIC_ACQUIRE) >> 32) != step_id)
        return false;
    return true;
  };
  if (e->sync_mode == 1) {
    // Poll the records the faces blocks write into pinned memory: lower wake-up
    // latency than a blocking stream synchronise.  Bounded spin, then fall back
    // to the stream so that device faults still surface as errors.
    // (the stream is asked only after 2 ms without the record, then every 2 ms: a
    // hipStreamQuery on a busy stream puts a system-scope barrier packet into the queue --
    // one per step when it was asked every 4096 spins, sitting between the step's last
    // launch and the next step's first: a full cache release + acquire the next launch
    // waited behind, 3 us per step where launches are queued ahead; profiles/r06_turn_around.txt)
    bool done = false;
    long long t_first = 0, t_query = 0;
    for (long spin = 0; !done; ++spin) {
      done = arrived();
      if (!done && (spin & 0xfff) == 0xfff) {
        const long long now = steady_ns();
        if (t_first == 0) t_first = t_query = now;
        if (now - t_first > 10000000000LL) break;  // 10 s: let the stream say what happened
        if (now - t_query > 2000000) {
          t_query = now;
          if (hipStreamQuery(e->stream) == hipSuccess)
            done = true;  // stream drained: the records must be there (or the kernel died)
        }
      }
    }
    if (!done) HIP_TRY(hipStreamSynchronize(e->stream));
  } else {
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (!arrived()) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!arrived()) return fail(FFN_ERR_HIP, "step %u did not complete", step_id);
  }
  if (n == 1) e->t_arrived_ns = steady_ns();
  {
    uint32_t* out = reinterpret_cast<uint32_t*>(results);
    for (int k = 0; k < n * kPubWords; ++k)
      out[k] = (uint32_t)__atomic_load_n(&h_pub[k], __ATOMIC_RELAXED);
  }
  for (int k = 0; k < n; ++k)
    if (results[k].range_error == 2) {
      if (spec_mismatch) {
        *spec_mismatch = true;
        return FFN_OK;
      }
      return fail(FFN_ERR_STATE,
                  "the speculative conv0_a launch of step %u chose another position "
                  "than the segment loop; the step changed nothing (speculate 0 "
                  "turns the launches off)", step_id);
    }
  bool any_void = false;
  for (int k = 0; k < n; ++k) any_void = any_void || results[k].range_error != 0;
  if (any_void) {
    bool timed_out = false;  // the cause is per step: two steps may be in flight
    for (int k = 0; k < n; ++k) timed_out = timed_out || results[k].range_error == 3;
    const int frc = flow_voided(e, timed_out);
    if (frc) return frc;
  } else {
    EngineLock lock_(e);
    if (e->slot_resident[slot]) e->flow_strikes = 0;
  }
  for (int k = 0; k < n; ++k)
    if (results[k].range_error)
      return fail(FFN_ERR_RANGE,
                  "an activation left the fp16 range (conv_variant >= 6): the "
                  "step changed nothing; set conv_variant -1 and repeat it");
  return FFN_OK;
}
}  // namespace

int ffn_canvas_step(ffn_engine* e, int n, ffn_canvas* const* canvases,
                    const ffn_step_request* requests,
                    const ffn_step_params* params, ffn_step_result* results) {
  if (!results) return fail(FFN_ERR_ARG, "null argument");
  uint32_t ticket = 0;
  int rc = ffn_canvas_step_submit(e, n, canvases, requests, params, &ticket);
  if (rc) return rc;
  bool mismatch = false;
  rc = step_wait_impl(e, ticket, results, &mismatch);
  if (rc || !mismatch) return rc;
  // The device's choice and the loop's differ (not expected: both apply
  // Canvas.is_valid_pos to the same values).  Nothing was pasted: the step is
  // made again, conv0_a included.
  {
    EngineLock lock_(e);
    e->stat_spec_mismatch += 1;
    drop_spec(e);
  }
  rc = ffn_canvas_step_submit(e, n, canvases, requests, params, &ticket);
  if (rc) return rc;
  return step_wait_impl(e, ticket, results, nullptr);
}

namespace {
// the HIP canvas as the device of the host loop
struct HipLoopDevice {
  ffn_canvas* c;
  int step(const ffn_step_request& req, const ffn_step_params& params,
           ffn_step_result* res) {
    ffn_canvas* one = c;
    return ffn_canvas_step(c->engine, 1, &one, &req, &params, res);
  }
  int read_point(const int32_t pos[3], float* seed, int32_t* seg) {
    return ffn_canvas_read_points(c, 1, pos, seed, seg);
  }
  // the positions the loop expects to pop after the step it is about to make
  void hint_next(int n, const int32_t (*pos)[3]) {
    EngineLock lock_(c->engine);
    c->hint_from_loop = true;
    c->hint_n = n < kSpecMax ? n : kSpecMax;
    for (int j = 0; j < c->hint_n; ++j)
      for (int a = 0; a < 3; ++a) c->hint_pos[j][a] = pos[j][a];
  }
};
}  // namespace

namespace {
// The step ffn_canvas_segment_many_carry left in flight: wait for it and hand its
// results to the loops of its canvases.  `only`: nothing to do unless that canvas
// is one of them.  Callers hold neither `mu` nor `many_mu`.
int resolve_many_carry_locked(ffn_engine* e) {
  if (!e->many_carry.active) return FFN_OK;
  HipLoopDevice dev{e->many_canvases.empty() ? nullptr : e->many_canvases[0]};
  auto wait = [&](ffn_step_result* res) {
    return step_wait_impl(e, e->many_ticket, res, nullptr);
  };
  const int rc = ffn_host::resolve_carry(&e->many_carry, dev, wait);
  e->many_canvases.clear();
  return rc;
}
int resolve_many_carry(ffn_engine* e, const ffn_canvas* only) {
  std::lock_guard<std::mutex> guard(e->many_mu);
  if (only && std::find(e->many_canvases.begin(), e->many_canvases.end(), only) ==
                  e->many_canvases.end())
    return FFN_OK;
  return resolve_many_carry_locked(e);
}
}  // namespace

int ffn_canvas_segment_at(ffn_canvas* c, const int32_t start[3],
                          const ffn_segment_params* p, int resume,
                          ffn_segment_result* out) {
  if (!c || !start || !p || !out) return fail(FFN_ERR_ARG, "null argument");
  if (!c->engine) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  if (int rc = resolve_many_carry(c->engine, c)) return rc;
  if (p->prefetch < 0 || p->prefetch > FFN_MAX_CANDIDATES)
    return fail(FFN_ERR_ARG, "prefetch must be 0..%d", FFN_MAX_CANDIDATES);
  if (p->shape_zyx[0] != c->cz || p->shape_zyx[1] != c->cy ||
      p->shape_zyx[2] != c->cx)
    return fail(FFN_ERR_ARG, "shape_zyx does not match the canvas");
  for (int a = 0; a < 3; ++a)
    if (p->deltas_zyx[a] < 0 || p->margin_zyx[a] < 0)
      return fail(FFN_ERR_ARG, "negative deltas / margin");
  if (resume && !c->loop.active)
    return fail(FFN_ERR_STATE, "no segment to resume");
  HipLoopDevice dev{c};
  ffn_host::SegmentLoop<HipLoopDevice> loop(dev, c->loop, *p);
  return loop.run(start, resume, out);
}

int ffn_canvas_segment_many(ffn_engine* e, int n, ffn_canvas* const* canvases,
                            const int32_t (*starts)[3],
                            const ffn_segment_params* params, const int32_t* resume,
                            ffn_segment_result* results, int32_t* finished) {
  return ffn_canvas_segment_many_carry(e, n, canvases, starts, params, resume, results,
                                       finished, 0);
}

int ffn_canvas_segment_many_carry(ffn_engine* e, int n, ffn_canvas* const* canvases,
                                  const int32_t (*starts)[3],
                                  const ffn_segment_params* params,
                                  const int32_t* resume, ffn_segment_result* results,
                                  int32_t* finished, int32_t carry) {
  if (!e || !canvases || !starts || !params || !resume || !results || !finished)
    return fail(FFN_ERR_ARG, "null argument");
  if (n < 1 || n > e->max_batch)
    return fail(FFN_ERR_ARG, "batch %d outside [1, %d]", n, e->max_batch);
  std::vector<HipLoopDevice> devs(n);
  std::vector<ffn_host::SegmentState*> states(n);
  for (int k = 0; k < n; ++k) {
    ffn_canvas* c = canvases[k];
    if (!c) return fail(FFN_ERR_ARG, "null canvas");
    if (c->engine != e) return fail(FFN_ERR_ARG, "canvas %d is not of this engine", k);
    for (int k2 = 0; k2 < k; ++k2)
      if (canvases[k2] == c) return fail(FFN_ERR_ARG, "canvas appears twice");
    const ffn_segment_params& p = params[k];
    if (p.prefetch < 0 || p.prefetch > FFN_MAX_CANDIDATES)
      return fail(FFN_ERR_ARG, "prefetch must be 0..%d", FFN_MAX_CANDIDATES);
    if (p.shape_zyx[0] != c->cz || p.shape_zyx[1] != c->cy || p.shape_zyx[2] != c->cx)
      return fail(FFN_ERR_ARG, "shape_zyx does not match canvas %d", k);
    for (int a = 0; a < 3; ++a)
      if (p.deltas_zyx[a] < 0 || p.margin_zyx[a] < 0)
        return fail(FFN_ERR_ARG, "negative deltas / margin");
    if (resume[k] && !c->loop.active)
      return fail(FFN_ERR_STATE, "canvas %d: no segment to resume", k);
    if (std::memcmp(&p.step, &params[0].step, sizeof(ffn_step_params)) != 0)
      return fail(FFN_ERR_ARG, "canvas %d: step parameters differ from canvas 0's", k);
    devs[k].c = c;
    states[k] = &c->loop;
  }
  std::vector<ffn_canvas*> batch(n);
  auto batch_step = [&](int nb, const int* idx, const ffn_step_request* reqs,
                        const ffn_step_params& sp, ffn_step_result* res) {
    for (int b = 0; b < nb; ++b) batch[b] = canvases[idx[b]];
    return ffn_canvas_step(e, nb, batch.data(), reqs, &sp, res);
  };
  // a step left in flight by the last call (of whichever kind) comes first; with
  // `carry` this call may leave one itself
  std::unique_lock<std::mutex> many_guard(e->many_mu);
  if (!carry) {
    const int rc = resolve_many_carry_locked(e);
    many_guard.unlock();
    if (rc) return rc;
    return ffn_host::segment_many(n, devs.data(), states.data(), starts, params,
                                  resume, results, finished, batch_step);
  }
  auto submit = [&](int nb, const int* idx, const ffn_step_request* reqs,
                    const ffn_step_params& sp) {
    for (int b = 0; b < nb; ++b) batch[b] = canvases[idx[b]];
    const int rc = ffn_canvas_step_submit(e, nb, batch.data(), reqs, &sp,
                                          &e->many_ticket);
    if (rc == FFN_OK) {
      e->many_canvases.assign(batch.begin(), batch.begin() + nb);
      e->stat_many_carried += 1;
    }
    return rc;
  };
  auto wait = [&](ffn_step_result* res) {
    const int rc = step_wait_impl(e, e->many_ticket, res, nullptr);
    e->many_canvases.clear();
    return rc;
  };
  return ffn_host::segment_many(n, devs.data(), states.data(), starts, params, resume,
                                results, finished, batch_step, &e->many_carry, submit,
                                wait);
}

int ffn_canvas_segment_history(ffn_canvas* c, size_t first, size_t n,
                               int32_t* pos, uint32_t* deleted, size_t* total) {
  if (!c) return fail(FFN_ERR_ARG, "null canvas");
  if (c->engine)
    if (int rc = resolve_many_carry(c->engine, c)) return rc;
  const size_t have = c->loop.history_deleted.size();
  if (total) *total = have;
  if (n == 0) return FFN_OK;
  if (first > have || n > have - first)
    return fail(FFN_ERR_ARG, "history range [%zu, +%zu) of %zu", first, n, have);
  if (pos) std::memcpy(pos, c->loop.history.data() + 3 * first, 12 * n);
  if (deleted) std::memcpy(deleted, c->loop.history_deleted.data() + first, 4 * n);
  return FFN_OK;
}

int ffn_canvas_read_points(ffn_canvas* c, int n, const int32_t* pos,
                           float* seed_out, int32_t* seg_out) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !pos || !seed_out || !seg_out) return fail(FFN_ERR_ARG, "null argument");
  if (n < 1) return FFN_OK;
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c, true));
  const size_t pb = sizeof(int32_t) * 3 * n;
  const size_t pbr = (pb + 15) & ~(size_t)15;
  int rc = ensure_scratch(e, pbr + 8 * (size_t)n);
  if (rc) return rc;
  char* hs = static_cast<char*>(e->h_scratch);
  char* ds = static_cast<char*>(e->d_scratch);
  std::memcpy(hs, pos, pb);
  HIP_TRY(hipMemcpyAsync(ds, hs, pb, hipMemcpyHostToDevice, e->ustream));
  float* d_seed = reinterpret_cast<float*>(ds + pbr);
  int32_t* d_seg = reinterpret_cast<int32_t*>(ds + pbr + 4 * (size_t)n);
  hipLaunchKernelGGL(points_read_kernel, dim3((n + 63) / 64), dim3(64), 0,
                     e->ustream, c->seed, c->seg, c->cz, c->cy, c->cx, n,
                     reinterpret_cast<const int32_t*>(ds), d_seed, d_seg);
  HIP_TRY(hipMemcpyAsync(hs + pbr, ds + pbr, 8 * (size_t)n, hipMemcpyDeviceToHost,
                         e->ustream));
  HIP_TRY(lock_.wait(e, c));
  std::memcpy(seed_out, hs + pbr, 4 * (size_t)n);
  std::memcpy(seg_out, hs + pbr + 4 * (size_t)n, 4 * (size_t)n);
  return FFN_OK;
}

int ffn_canvas_write_seg_points(ffn_canvas* c, int n, const int32_t* pos,
                                const int32_t* values) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !pos || !values) return fail(FFN_ERR_ARG, "null argument");
  if (n < 1) return FFN_OK;
  for (int k = 0; k < n; ++k)
    if (pos[3 * k] < 0 || pos[3 * k] >= c->cz || pos[3 * k + 1] < 0 ||
        pos[3 * k + 1] >= c->cy || pos[3 * k + 2] < 0 || pos[3 * k + 2] >= c->cx)
      return fail(FFN_ERR_ARG, "point %d outside the canvas", k);
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c));
  if (n == 1) {  // (the -1 marker of a rejected seed: nothing to wait for)
    const size_t ci = ((size_t)pos[0] * c->cy + pos[1]) * c->cx + pos[2];
    hipLaunchKernelGGL(set_seg_point_kernel, dim3(1), dim3(1), 0, e->ustream, c->seg,
                       ci, values[0]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(lock_.end(e, c));
    return FFN_OK;
  }
  const size_t pb = sizeof(int32_t) * 3 * n;
  const size_t pbr = (pb + 15) & ~(size_t)15;
  int rc = ensure_scratch(e, pbr + 4 * (size_t)n);
  if (rc) return rc;
  char* hs = static_cast<char*>(e->h_scratch);
  char* ds = static_cast<char*>(e->d_scratch);
  std::memcpy(hs, pos, pb);
  std::memcpy(hs + pbr, values, 4 * (size_t)n);
  HIP_TRY(hipMemcpyAsync(ds, hs, pbr + 4 * (size_t)n, hipMemcpyHostToDevice,
                         e->ustream));
  hipLaunchKernelGGL(points_write_seg_kernel, dim3((n + 63) / 64), dim3(64), 0,
                     e->ustream, c->seg, c->cy, c->cx, n,
                     reinterpret_cast<const int32_t*>(ds),
                     reinterpret_cast<const int32_t*>(ds + pbr));
  HIP_TRY(lock_.wait(e, c));
  return FFN_OK;
}

int ffn_canvas_any_segmented(ffn_canvas* c, const int32_t lo[3],
                             const int32_t hi[3], int32_t* out) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !lo || !hi || !out) return fail(FFN_ERR_ARG, "null argument");
  // numpy slicing clips to the array bounds (inference.py:575-578)
  int32_t l[3], h[3];
  const int dims[3] = {c->cz, c->cy, c->cx};
  for (int k = 0; k < 3; ++k) {
    l[k] = std::max(lo[k], 0);
    h[k] = std::min(hi[k], dims[k]);
    if (h[k] <= l[k]) {
      *out = 0;
      return FFN_OK;
    }
  }
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c));
  int rc = ensure_scratch(e, 16);
  if (rc) return rc;
  long total;
  Box b = make_box(c, l, h, &total);
  HIP_TRY(hipMemsetAsync(e->d_scratch, 0, 4, e->ustream));
  hipLaunchKernelGGL(any_segmented_kernel, dim3(grid_for(total)), dim3(256), 0,
                     e->ustream, c->seg, b, total,
                     static_cast<int32_t*>(e->d_scratch));
  HIP_TRY(hipMemcpyAsync(e->h_scratch, e->d_scratch, 4, hipMemcpyDeviceToHost,
                         e->ustream));
  HIP_TRY(lock_.wait(e, c));
  *out = *static_cast<int32_t*>(e->h_scratch);
  return FFN_OK;
}

int ffn_canvas_commit_count(ffn_canvas* c, const int32_t lo[3],
                            const int32_t hi[3], float segment_threshold,
                            int32_t max_existing_id, ffn_commit_counts* counts,
                            int32_t max_overlaps, int32_t* overlap_ids,
                            int64_t* overlap_counts) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !lo || !hi || !counts) return fail(FFN_ERR_ARG, "null argument");
  int rc = check_box(c, lo, hi);
  if (rc) return rc;
  if (max_existing_id < 0) max_existing_id = 0;
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c));
  const size_t hist_n = (size_t)max_existing_id + 1;
  const size_t bytes = 16 + hist_n * sizeof(unsigned);
  rc = ensure_scratch(e, bytes);
  if (rc) return rc;
  long total;
  Box b = make_box(c, lo, hi, &total);
  HIP_TRY(hipMemsetAsync(e->d_scratch, 0, bytes, e->ustream));
  auto* d_counts = static_cast<unsigned long long*>(e->d_scratch);
  auto* d_hist = reinterpret_cast<unsigned*>(static_cast<char*>(e->d_scratch) + 16);
  if (total > 0)
    hipLaunchKernelGGL(commit_count_kernel, dim3(grid_for(total)), dim3(256), 0,
                       e->ustream, c->seed, c->seg, b, total, segment_threshold,
                       max_existing_id, d_counts, d_hist);
  HIP_TRY(hipMemcpyAsync(e->h_scratch, e->d_scratch, bytes, hipMemcpyDeviceToHost,
                         e->ustream));
  HIP_TRY(lock_.wait(e, c));
  const auto* hc = static_cast<const unsigned long long*>(e->h_scratch);
  const auto* hh = reinterpret_cast<const unsigned*>(
      static_cast<const char*>(e->h_scratch) + 16);
  counts->raw_segmented_voxels = (int64_t)hc[0];
  counts->actual_segmented_voxels = (int64_t)hc[1];
  int nover = 0;
  for (size_t id = 1; id < hist_n; ++id) {
    if (hh[id]) {
      if (nover < max_overlaps && overlap_ids && overlap_counts) {
        overlap_ids[nover] = (int32_t)id;
        overlap_counts[nover] = (int64_t)hh[id];
      }
      ++nover;
    }
  }
  counts->num_overlapped_ids = nover;
  return FFN_OK;
}

int ffn_canvas_commit_assign(ffn_canvas* c, const int32_t lo[3],
                             const int32_t hi[3], float segment_threshold,
                             int32_t segment_id) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !lo || !hi) return fail(FFN_ERR_ARG, "null argument");
  int rc = check_box(c, lo, hi);
  if (rc) return rc;
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(lock_.begin(e, c));
  long total;
  Box b = make_box(c, lo, hi, &total);
  if (total > 0)
    hipLaunchKernelGGL(commit_assign_kernel, dim3(grid_for(total)), dim3(256), 0,
                       e->ustream, c->seed, c->seg, b, total, segment_threshold,
                       segment_id);
  HIP_TRY(hipGetLastError());
  HIP_TRY(lock_.end(e, c));
  return FFN_OK;
}

int ffn_canvas_segment_turn(ffn_canvas* c, const ffn_turn_request* rq,
                            const int32_t* cand, ffn_turn_result* out,
                            int32_t max_overlaps, int32_t* overlap_ids,
                            int64_t* overlap_counts, int32_t* cand_flags,
                            float* cand_seed, int32_t* cand_seg) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !rq || !out) return fail(FFN_ERR_ARG, "null argument");
  const int n = rq->num_candidates;
  if (n < 0 || (n > 0 && (!cand || !cand_flags || !cand_seed || !cand_seg)))
    return fail(FFN_ERR_ARG, "candidate arrays");
  int rc = FFN_OK;
  if (rq->do_commit && (rc = check_box(c, rq->lo, rq->hi))) return rc;
  auto inside = [&](const int32_t* p) {
    return p[0] >= 0 && p[0] < c->cz && p[1] >= 0 && p[1] < c->cy && p[2] >= 0 &&
           p[2] < c->cx;
  };
  if (rq->mark_mode && !inside(rq->mark_pos))
    return fail(FFN_ERR_ARG, "marker outside the canvas");
  for (int k = 0; k < n; ++k)
    if (!inside(cand + 3 * k)) return fail(FFN_ERR_ARG, "candidate %d outside the canvas", k);
  for (int a = 0; a < 3; ++a)
    if (rq->min_boundary_dist[a] < 0) return fail(FFN_ERR_ARG, "min_boundary_dist");
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  const long long t_turn0 = steady_ns();
  HIP_TRY(lock_.begin(e, c));
  // scratch: [record][histogram][flags, seeds, segs: n each][candidates: 3 n]
  const int32_t max_id = rq->do_commit ? std::max(rq->max_existing_id, 0) : 0;
  const size_t hist_n = rq->do_commit ? (size_t)max_id + 1 : 0;
  const size_t o_hist = sizeof(TurnRecord);
  const size_t o_flag = (o_hist + hist_n * 4 + 15) & ~(size_t)15;
  const size_t o_cand = o_flag + 12 * (size_t)n;
  const size_t bytes = o_cand + 12 * (size_t)n;
  rc = ensure_scratch(e, bytes);
  if (rc) return rc;
  char* ds = static_cast<char*>(e->d_scratch);
  char* hs = static_cast<char*>(e->h_scratch);
  auto* d_rec = reinterpret_cast<TurnRecord*>(ds);
  auto* d_hist = reinterpret_cast<unsigned*>(ds + o_hist);
  auto* d_flag = reinterpret_cast<int*>(ds + o_flag);
  auto* d_cseed = reinterpret_cast<float*>(ds + o_flag + 4 * (size_t)n);
  auto* d_cseg = reinterpret_cast<int32_t*>(ds + o_flag + 8 * (size_t)n);
  auto* d_cand = reinterpret_cast<int32_t*>(ds + o_cand);
  // (chosen = -1 until the pick kernel has run: all-ones bytes)
  HIP_TRY(hipMemsetAsync(ds, 0, o_flag, e->ustream));
  HIP_TRY(hipMemsetAsync(&d_rec->chosen, 0xff, 4, e->ustream));
  if (n > 0) {
    std::memcpy(hs + o_cand, cand, 12 * (size_t)n);
    HIP_TRY(hipMemcpyAsync(d_cand, hs + o_cand, 12 * (size_t)n, hipMemcpyHostToDevice,
                           e->ustream));
  }
  long total = 0;
  Box b{};
  if (rq->do_commit) {
    b = make_box(c, rq->lo, rq->hi, &total);
    if (total > 0)
      hipLaunchKernelGGL(commit_count_kernel, dim3(grid_for(total)), dim3(256), 0,
                         e->ustream, c->seed, c->seg, b, total, rq->segment_threshold,
                         max_id, d_rec->counts, d_hist);
  }
  if (rq->do_commit || rq->mark_mode) {
    const long mark_ci =
        rq->mark_mode ? ((long)rq->mark_pos[0] * c->cy + rq->mark_pos[1]) * c->cx +
                            rq->mark_pos[2]
                      : 0;
    hipLaunchKernelGGL(turn_commit_kernel, dim3(total > 0 ? grid_for(total) : 1),
                       dim3(256), 0, e->ustream, c->seed, c->seg, b, total,
                       rq->segment_threshold, rq->segment_id,
                       (long long)rq->min_segment_size, d_rec, mark_ci, rq->mark_mode);
  }
  long dirty_total = 0;
  if (n > 0) {
    hipLaunchKernelGGL(turn_eval_kernel, dim3(n), dim3(64), 0, e->ustream, c->seed,
                       c->seg, c->cz, c->cy, c->cx, d_cand, rq->min_boundary_dist[0],
                       rq->min_boundary_dist[1], rq->min_boundary_dist[2], d_flag,
                       d_cseed, d_cseg);
    hipLaunchKernelGGL(turn_pick_kernel, dim3(1), dim3(64), 0, e->ustream, c->seg,
                       c->cy, c->cx, d_cand, n, d_flag, d_rec);
    if (rq->do_init) {
      if (c->dirty_lo[0] < c->dirty_hi[0]) {
        Box db = make_box(c, c->dirty_lo, c->dirty_hi, &dirty_total);
        const int linear = (size_t)dirty_total * 2 >= c->nvox;
        if (dirty_total > 0)
          hipLaunchKernelGGL(turn_clear_kernel,
                             dim3(linear ? 2048 : grid_for(dirty_total)), dim3(256), 0,
                             e->ustream, reinterpret_cast<uint32_t*>(c->seed), db,
                             dirty_total, linear, c->nvox, d_rec);
      }
      hipLaunchKernelGGL(turn_seed_kernel, dim3(1), dim3(1), 0, e->ustream, c->seed,
                         c->cy, c->cx, d_cand, rq->init_value, d_rec);
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hs, ds, o_cand, hipMemcpyDeviceToHost, e->ustream));
  const long long t_turn1 = steady_ns();
  HIP_TRY(lock_.wait(e, c));
  e->stat_segturn_queue_ns += t_turn1 - t_turn0;
  e->stat_segturn_wait_ns += steady_ns() - t_turn1;
  e->stat_segturn_calls += 1;
  const auto* hr = reinterpret_cast<const TurnRecord*>(hs);
  std::memset(out, 0, sizeof(*out));
  out->counts.raw_segmented_voxels = (int64_t)hr->counts[0];
  out->counts.actual_segmented_voxels = (int64_t)hr->counts[1];
  out->committed = hr->committed;
  out->chosen = n > 0 ? hr->chosen : -1;
  const auto* hh = reinterpret_cast<const unsigned*>(hs + o_hist);
  int nover = 0;
  for (size_t id = 1; id < hist_n; ++id) {
    if (hh[id]) {
      if (nover < max_overlaps && overlap_ids && overlap_counts) {
        overlap_ids[nover] = (int32_t)id;
        overlap_counts[nover] = (int64_t)hh[id];
      }
      ++nover;
    }
  }
  out->counts.num_overlapped_ids = nover;
  if (n > 0) {
    std::memcpy(cand_flags, hs + o_flag, 4 * (size_t)n);
    std::memcpy(cand_seed, hs + o_flag + 4 * (size_t)n, 4 * (size_t)n);
    std::memcpy(cand_seg, hs + o_flag + 8 * (size_t)n, 4 * (size_t)n);
    if (rq->do_init && out->chosen >= 0) {
      // as ffn_canvas_init_seed: the volume is clean but for the seed point
      const int32_t* p = cand + 3 * out->chosen;
      const int lo[3] = {p[0], p[1], p[2]};
      const int hi[3] = {p[0] + 1, p[1] + 1, p[2] + 1};
      c->dirty_lo[0] = c->dirty_hi[0] = 0;
      c->mark_dirty(lo, hi);
    }
  }
  return FFN_OK;
}

}  // extern "C"

namespace {

template <typename T>
int box_read(ffn_canvas* c, const T* vol, const int32_t lo[3], const int32_t hi[3],
             T* dst) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !lo || !hi || !dst) return fail(FFN_ERR_ARG, "null argument");
  int rc = check_box(c, lo, hi);
  if (rc) return rc;
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  long total;
  Box b = make_box(c, lo, hi, &total);
  if (total == 0) return FFN_OK;
  if (total == (long)c->nvox) {  // whole volume: straight (blocking) copy
    HIP_TRY(canvas_quiesce(e, c));
    HIP_TRY(hipMemcpy(dst, vol, sizeof(T) * total, hipMemcpyDeviceToHost));
    return FFN_OK;
  }
  HIP_TRY(lock_.begin(e, c));
  rc = ensure_scratch(e, sizeof(T) * total);
  if (rc) return rc;
  hipLaunchKernelGGL((box_read_kernel<T>), dim3(grid_for(total)), dim3(256), 0,
                     e->ustream, vol, b, total, static_cast<T*>(e->d_scratch));
  HIP_TRY(hipMemcpyAsync(e->h_scratch, e->d_scratch, sizeof(T) * total,
                         hipMemcpyDeviceToHost, e->ustream));
  HIP_TRY(lock_.wait(e, c));
  std::memcpy(dst, e->h_scratch, sizeof(T) * total);
  return FFN_OK;
}

template <typename T>
int box_write(ffn_canvas* c, T* vol, const int32_t lo[3], const int32_t hi[3],
              const T* src) {
  UtilLock lock_(c ? c->engine : nullptr);
  if (!c || !lo || !hi || !src) return fail(FFN_ERR_ARG, "null argument");
  int rc = check_box(c, lo, hi);
  if (rc) return rc;
  ffn_engine* e = c->engine;
  if (!e) return fail(FFN_ERR_STATE, "canvas outlived its engine");
  HIP_TRY(hipSetDevice(e->device));
  long total;
  Box b = make_box(c, lo, hi, &total);
  if (total == 0) return FFN_OK;
  if (total == (long)c->nvox) {  // whole volume: straight (blocking) copy
    HIP_TRY(canvas_quiesce(e, c));
    HIP_TRY(hipMemcpy(vol, src, sizeof(T) * total, hipMemcpyHostToDevice));
    return FFN_OK;
  }
  HIP_TRY(lock_.begin(e, c));
  rc = ensure_scratch(e, sizeof(T) * total);
  if (rc) return rc;
  std::memcpy(e->h_scratch, src, sizeof(T) * total);
  HIP_TRY(hipMemcpyAsync(e->d_scratch, e->h_scratch, sizeof(T) * total,
                         hipMemcpyHostToDevice, e->ustream));
  hipLaunchKernelGGL((box_write_kernel<T>), dim3(grid_for(total)), dim3(256), 0,
                     e->ustream, vol, b, total,
                     static_cast<const T*>(e->d_scratch));
  HIP_TRY(lock_.wait(e, c));
  return FFN_OK;
}

}  // namespace

extern "C" {

int ffn_canvas_read_seed(ffn_canvas* c, const int32_t lo[3], const int32_t hi[3],
                         float* dst) {
  return box_read<float>(c, c ? c->seed : nullptr, lo, hi, dst);
}

int ffn_canvas_read_segmentation(ffn_canvas* c, const int32_t lo[3],
                                 const int32_t hi[3], int32_t* dst) {
  return box_read<int32_t>(c, c ? c->seg : nullptr, lo, hi, dst);
}

int ffn_canvas_write_seed(ffn_canvas* c, const int32_t lo[3], const int32_t hi[3],
                          const float* src) {
  if (c && lo && hi) c->mark_dirty(lo, hi);
  return box_write<float>(c, c ? c->seed : nullptr, lo, hi, src);
}

int ffn_canvas_write_segmentation(ffn_canvas* c, const int32_t lo[3],
                                  const int32_t hi[3], const int32_t* src) {
  return box_write<int32_t>(c, c ? c->seg : nullptr, lo, hi, src);
}

}  // extern "C"
