/*
 * libffn_hip.so -- C-ABI of the MI355X (gfx950) flood-filling inference engine.
 *
 * This is the drop-in boundary for the ONE hot path this repository replaces:
 * the FFN field-of-view loop of google/ffn.  Every entry point below names the
 * reference interface (file:line, relative to the google/ffn checkout) whose
 * work it takes over.  The reference is pure Python; its FFI for this path is a
 * ctypes binding (see INTEGRATION.md for the stub a maintainer would add to
 * ffn/inference/executor.py).
 *
 * Conventions: plain C types only; every function returns 0 on success or a
 * negative FFN_ERR_* code (no exceptions cross the ABI); the caller owns all
 * host buffers, the library owns all device memory; one HIP stream per engine;
 * calls on one engine (and on canvases created from it) must be serialised by
 * the caller (the executor's server thread); coordinates are (z, y, x).
 */
#ifndef FFN_HIP_H_
#define FFN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFN_OK 0
#define FFN_ERR_ARG (-1)     /* invalid argument / unsupported geometry      */
#define FFN_ERR_HIP (-2)     /* HIP runtime failure (see ffn_last_error)     */
#define FFN_ERR_STATE (-3)   /* call order violated (e.g. weights not set)   */
#define FFN_ERR_RANGE (-4)   /* conv_variant >= 6 only: an operand left the fp16
                                range; the step changed nothing -- switch to
                                conv_variant -1 (exact f32) and repeat it       */

#define FFN_ERR_FLOW (-5)    /* option flow = 2 only: the resident conv launch
                                timed out waiting for one of its own workgroups
                                (a shared or partitioned GPU); the step changed
                                nothing -- repeat it: the repeat runs one launch
                                per conv (same arithmetic), and after 3 such steps
                                in a row the engine sets flow = 0 by itself      */

#define FFN_MAX_CANDIDATES 16

typedef struct ffn_engine ffn_engine;
typedef struct ffn_canvas ffn_canvas;

/* Thresholds of one FoV step, already in logit space, exactly the f32-rounded
 * values Canvas.__init__ stores (ffn/inference/inference.py:189-195). */
typedef struct ffn_step_params {
  float pad_value;             /* NaN -> pad substitution  (inference.py:406-407) */
  float move_threshold;        /* disco mean test + counters (inference.py:429)   */
  float disco_seed_threshold;  /* < 0 disables the disco bias (inference.py:416)  */
  float deleted_threshold;     /* keep_history (inference.py:420-423): count voxels
                                  with old seed >= this and new logit < 0; NaN =
                                  do not count                                    */
} ffn_step_params;

/* One FoV step request (one entry per canvas in a batched call). */
typedef struct ffn_step_request {
  int32_t pos[3];        /* FoV centre (z, y, x)                               */
  int32_t start_pos[3];  /* segment origin: its logit is returned for the
                            "seed got too weak" test (inference.py:503-505)    */
  int32_t num_candidates;                      /* <= FFN_MAX_CANDIDATES        */
  int32_t candidates[FFN_MAX_CANDIDATES][3];   /* queue-head positions whose
                            post-step seed logit / segment id are wanted
                            (Canvas.is_valid_pos, inference.py:325,341)        */
} ffn_step_request;

/* Everything the Python movement policy needs from one FoV step. */
typedef struct ffn_step_result {
  float   face_score[6];  /* max over each face of the +-delta cuboid, order
                             (z-,z+,y-,y+,x-,x+)  (movement.py:67-87)          */
  int32_t face_index[6];  /* first-occurrence C-order argmax inside the face   */
  int32_t face_seg[6];    /* segmentation[] at each face maximum (the validity
                             test of a freshly queued move, inference.py:341)  */
  float   start_logit;    /* seed[start_pos] after the paste                   */
  uint32_t num_above_move;/* #(logits >= move_threshold) before disco          */
  int32_t disco_applied;  /* 1 if the disco mask was applied                   */
  float   cand_seed[FFN_MAX_CANDIDATES];  /* seed[candidate] after the paste   */
  int32_t cand_seg[FFN_MAX_CANDIDATES];   /* segmentation[candidate]           */
  uint32_t num_deleted;   /* history_deleted entry of this step (see
                             ffn_step_params.deleted_threshold), else 0        */
  int32_t range_error;    /* conv_variant >= 6: 1 = an operand left the fp16 range,
                             the step was NOT pasted (FFN_ERR_RANGE); 2 = the
                             step's conv0_a was a launch made ahead for another
                             position (option "speculate"): NOT pasted either,
                             ffn_canvas_step repeats such a step itself; 3 = the
                             resident conv launch gave up waiting for one of its
                             own workgroups: NOT pasted (FFN_ERR_FLOW)          */
} ffn_step_result;

/* Result of the per-segment commit reduction (inference.py:614-646). */
typedef struct ffn_commit_counts {
  int64_t raw_segmented_voxels;     /* sum(seed[sel] >= segment_threshold)      */
  int64_t actual_segmented_voxels;  /* ... and segmentation[sel] <= 0           */
  int32_t num_overlapped_ids;       /* distinct ids > 0 under the raw mask      */
} ffn_commit_counts;

/* Canvas.segment_at as ONE call: the FoV loop of a segment with the default
 * movement policy, run by the library's host code (ffn_canvas_segment_at). */
typedef struct ffn_segment_params {
  ffn_step_params step;       /* thresholds of every FoV step                   */
  double  score_threshold;    /* FaceMaxMovementPolicy.score_threshold
                                 (movement.py:176,214): logit(move_threshold)  */
  int32_t deltas_zyx[3];      /* policy deltas (model_info.deltas[::-1])        */
  int32_t margin_zyx[3];      /* input_image_size // 2 (inference.py:216)       */
  int32_t shape_zyx[3];       /* canvas shape, for the bounds test              */
  int32_t init_min_pos[3];    /* Canvas._min_pos / _max_pos after reset_state   */
  int32_t init_max_pos[3];    /*   (inference.py:288-310)                       */
  float   initial_start_logit;/* seed[start] if known (init_seed), else NaN     */
  int32_t prefetch;           /* queue-head candidates sent with a step (<= 16) */
  int32_t keep_history;       /* record positions (+ deleted counts)            */
  int64_t max_steps;          /* > 0: return after this many steps (resumable)  */
} ffn_segment_params;

typedef struct ffn_segment_result {
  int64_t num_steps;          /* FoV steps made by this call                    */
  int64_t skip_threshold;     /* counters of Canvas.is_valid_pos                */
  int64_t skip_invalid_pos;
  int64_t gate_rejects;       /* in-bounds entries refused on device state      */
  int64_t queue_len;          /* entries left in the policy queue               */
  int32_t seed_got_too_weak;  /* loop left through inference.py:503-505         */
  int32_t budget_exhausted;   /* max_steps reached; call again with resume = 1  */
  int32_t active;             /* the segment can be resumed                     */
  int32_t start_logit_known;
  float   start_logit;        /* seed[start] after the last step                */
  int32_t min_pos[3];         /* extents of the visited positions               */
  int32_t max_pos[3];
} ffn_segment_result;

/* ---- lifecycle -----------------------------------------------------------
 * Replaces model construction in Runner._init_tf_model
 * (ffn/inference/runner.py:116-163) and ConvStack3DFFNModel.__init__ /
 * define_tf_graph (ffn/training/models/convstack_3d.py:68-95).
 * features must be 32 (the MFMA kernel's N); depth >= 1; fov odd per axis. */
int ffn_engine_create(int device_id, const int32_t fov_zyx[3],
                      const int32_t deltas_zyx[3], int depth, int features,
                      int max_batch, ffn_engine** out);
void ffn_engine_destroy(ffn_engine* engine);

/* Replaces tf.train.Saver().restore (runner.py:98-111).  `blob` holds the TF
 * variables in graph order, each tensor exactly as TensorFlow stores it:
 *   conv0_a W[3][3][3][2][F] b[F], conv0_b W[3][3][3][F][F] b[F],
 *   conv{i}_a W b, conv{i}_b W b  (i = 1..depth-1), conv_lom W[F] b[1].     */
int ffn_engine_set_weights(ffn_engine* engine, const float* blob, size_t count);
size_t ffn_engine_weight_count(int depth, int features);

/* ---- stateless step == ExecutorClient.predict ----------------------------
 * Replaces BatchExecutor._schedule_batch -> session.run
 * (ffn/inference/executor.py:313-340): logits = seed + conv_stack(image,seed).
 * seed, image, logits_out: host arrays [n][fz][fy][fx] f32, n <= max_batch.   */
int ffn_predict(ffn_engine* engine, int n, const float* seed,
                const float* image, float* logits_out);

/* Conv stack only, on whatever FoVs are resident in the engine's staging
 * buffers (no host traffic) -- the kernel-only leg of bench.py. */
int ffn_forward_resident(ffn_engine* engine, int n, int repeats);

/* ---- device-resident canvas ---------------------------------------------
 * Replaces the numpy state of Canvas (inference.py:224-232): image f32
 * (already normalised as runner.py:383-385), seed f32 (NaN = unvisited),
 * segmentation i32, all kept in HBM for the whole subvolume. */
int ffn_canvas_create(ffn_engine* engine, const float* image_f32,
                      const int32_t shape_zyx[3], ffn_canvas** out);
/* The same from the RAW uint8 image: Runner.make_canvas' normalisation
 * (image.astype(np.float32) - image_mean) / image_stddev (runner.py:383-385)
 * moves into the FoV gather of the first conv (a 256-entry table made with the
 * same two f32 operations), the image stays 1 byte per voxel in HBM (1024^3:
 * 1.07 GB instead of 4.29 GB) and no normalised f32 copy is made on the host.
 * Every result is bit-identical to ffn_canvas_create on the normalised image. */
int ffn_canvas_create_u8(ffn_engine* engine, const uint8_t* image_u8,
                         const int32_t shape_zyx[3], float image_mean,
                         float image_stddev, ffn_canvas** out);
void ffn_canvas_destroy(ffn_canvas* canvas);

/* Canvas.init_seed (inference.py:443-450): seed[:] = NaN; seed[pos] = value. */
int ffn_canvas_init_seed(ffn_canvas* canvas, const int32_t pos[3], float value);

/* Canvas.update_at + FaceMaxMovementPolicy scoring for n canvases at once
 * (inference.py:386-441, movement.py:42-100): gather -> conv stack -> disco ->
 * paste-back -> 6-face argmax, entirely on the GPU. */
int ffn_canvas_step(ffn_engine* engine, int n, ffn_canvas* const* canvases,
                    const ffn_step_request* requests,
                    const ffn_step_params* params, ffn_step_result* results);

/* The same step split in two, so that the host work of one group of canvases
 * (queue bookkeeping in Python) overlaps the GPU work of another: submit
 * enqueues the whole step and returns at once; wait blocks until its results
 * are in host memory.  At most two steps may be in flight per engine and a
 * canvas may be in only one of them; steps execute in submission order.
 * ffn_canvas_step == submit + wait.  Other calls on the engine's canvases are
 * ordered behind the steps in flight (same stream). */
int ffn_canvas_step_submit(ffn_engine* engine, int n,
                           ffn_canvas* const* canvases,
                           const ffn_step_request* requests,
                           const ffn_step_params* params, uint32_t* ticket);
int ffn_canvas_step_wait(ffn_engine* engine, uint32_t ticket,
                         ffn_step_result* results);

/* Canvas.segment_at (inference.py:460-533) with FaceMaxMovementPolicy
 * (movement.py:166-222) and Canvas.is_valid_pos (inference.py:312-346), looped
 * inside the library: pop a position, test it, one ffn_canvas_step, queue the
 * face maxima -- until the queue is empty, the start logit falls below
 * move_threshold, or max_steps.  The seed must have been initialised
 * (ffn_canvas_init_seed) by the caller.  resume = 1 continues a segment that
 * returned with budget_exhausted or with an error (e.g. FFN_ERR_RANGE: switch
 * conv_variant and resume; the voided step is repeated).  The visited positions
 * are those of the Python loop, step for step. */
int ffn_canvas_segment_at(ffn_canvas* canvas, const int32_t start_zyx[3],
                          const ffn_segment_params* params, int resume,
                          ffn_segment_result* result);
/* Config C3 in the library (the reference's client threads + batching server
 * thread, ffn/inference/executor.py:266-340, with the per-canvas policy queues
 * here): the ffn_canvas_segment_at loops of n canvases of one engine advanced
 * TOGETHER, one batched ffn_canvas_step per round over the loops still running.
 * Returns as soon as at least one loop has ended (queue empty, seed too weak, or
 * its max_steps spent in this call): finished[k] = 1 marks it, and the caller
 * finishes that segment and gives the canvas its next one (resume[k] = 0); the
 * others continue with resume[k] = 1, in the next call or a later one.
 * results[k] counts from the START of canvas k's segment over all its calls
 * (budget_exhausted aside).  n <= max_batch, canvases distinct, every
 * params[k].step identical (one engine call = one set of step parameters).  On
 * an error (e.g. FFN_ERR_RANGE) nothing of the failed round was pasted and
 * every prepared position stays pending: deal with it, then call again with
 * resume = 1 for every canvas. */
int ffn_canvas_segment_many(ffn_engine* engine, int n, ffn_canvas* const* canvases,
                            const int32_t (*starts_zyx)[3],
                            const ffn_segment_params* params,
                            const int32_t* resume, ffn_segment_result* results,
                            int32_t* finished);

/* ... with the others' next step left IN FLIGHT when a loop ends (carry != 0):
 * the call queues the batched step the still-running loops have prepared and
 * returns without waiting for it, so that the caller's between-segment work on
 * the canvas that ended (ffn_canvas_segment_turn, its bookkeeping, the next
 * call's arguments) runs under that step instead of in front of it -- what a
 * second group of canvases on a second host thread otherwise provides
 * (executor.py:266-340: the reference's client threads).  The next
 * ffn_canvas_segment_many[_carry] call on the engine waits for the step first and
 * feeds its results to the loops it was made for; so does any entry point that
 * needs one of those canvases' loops earlier (ffn_canvas_segment_at,
 * ffn_canvas_segment_history, ffn_canvas_destroy).  results[k].num_steps counts
 * a carried step once its results are in.  One carried step per engine: meant
 * for ONE driving thread; max_steps budgets are per call and do not see the
 * carried step (callers with budgets pass carry = 0).  carry = 0:
 * ffn_canvas_segment_many. */
int ffn_canvas_segment_many_carry(ffn_engine* engine, int n,
                                  ffn_canvas* const* canvases,
                                  const int32_t (*starts_zyx)[3],
                                  const ffn_segment_params* params,
                                  const int32_t* resume,
                                  ffn_segment_result* results, int32_t* finished,
                                  int32_t carry);
/* keep_history: entries [first, first + n) of the current segment's history
 * (positions zyx, deleted-voxel counts); *total = entries recorded. */
int ffn_canvas_segment_history(ffn_canvas* canvas, size_t first, size_t n,
                               int32_t* pos_zyx, uint32_t* deleted,
                               size_t* total);

/* Point reads used by Canvas.is_valid_pos (inference.py:312-346). */
int ffn_canvas_read_points(ffn_canvas* canvas, int n, const int32_t* pos_zyx,
                           float* seed_out, int32_t* seg_out);
/* segmentation[pos] = value (the -1 "excluded" markers, inference.py:579,604). */
int ffn_canvas_write_seg_points(ffn_canvas* canvas, int n,
                                const int32_t* pos_zyx, const int32_t* values);
/* np.any(segmentation[lo:hi] > 0) -- min_boundary_dist test (inference.py:573-581). */
int ffn_canvas_any_segmented(ffn_canvas* canvas, const int32_t lo[3],
                             const int32_t hi[3], int32_t* out);

/* Segment commit (inference.py:614-646), in two calls so Python keeps the
 * accept/reject decision: count, then (if accepted) assign `segment_id`.
 * overlap_ids/overlap_counts receive up to `max_overlaps` (id, count) pairs in
 * ascending id order (np.unique semantics, ids > 0 only). */
int ffn_canvas_commit_count(ffn_canvas* canvas, const int32_t lo[3],
                            const int32_t hi[3], float segment_threshold,
                            int32_t max_existing_id, ffn_commit_counts* counts,
                            int32_t max_overlaps, int32_t* overlap_ids,
                            int64_t* overlap_counts);
int ffn_canvas_commit_assign(ffn_canvas* canvas, const int32_t lo[3],
                             const int32_t hi[3], float segment_threshold,
                             int32_t segment_id);

/* The between-segment turn of Canvas.segment_all (inference.py:573-660) as one
 * device-side sequence, one host wait at its end instead of one per question:
 *   1. do_commit: the commit reduction over [lo, hi) (ffn_canvas_commit_count);
 *      if actual_segmented_voxels >= min_segment_size the voxels get segment_id
 *      (inference.py:614-646) and *committed = 1;
 *   2. mark_mode 1: segmentation[mark_pos] = -1 if it is 0 ("weak seed",
 *      inference.py:600-603); 2: the same when step 1 did not commit ("too
 *      small", inference.py:632-636); 0: no marker;
 *   3. the next seeds of the policy, `candidates` in its order (all inside the
 *      canvas): a candidate whose segmentation is > 0 is skipped
 *      (Canvas.is_valid_pos, inference.py:341), one with an id > 0 within
 *      min_boundary_dist (clipped box, inference.py:575-581) is marked -1 and
 *      skipped; the first one that passes is *chosen* (-1: none of them);
 *   4. do_init: Canvas.init_seed(candidates[chosen], init_value)
 *      (inference.py:282-286) when there is one.
 * cand_flags[k]: 0 passed (k = chosen), 1 already segmented, 2 too close (marked),
 * 3 after the chosen one (untouched); cand_seed / cand_seg: the canvas values at
 * candidate k as step 3 saw them. */
typedef struct ffn_turn_request {
  int32_t do_commit;
  int32_t lo[3], hi[3];
  float   segment_threshold;
  int64_t min_segment_size;
  int32_t segment_id;
  int32_t max_existing_id;
  int32_t mark_mode;
  int32_t mark_pos[3];
  int32_t num_candidates;
  int32_t min_boundary_dist[3];   /* zyx */
  int32_t do_init;
  float   init_value;
} ffn_turn_request;

typedef struct ffn_turn_result {
  ffn_commit_counts counts;       /* zero without do_commit                     */
  int32_t committed;
  int32_t chosen;
} ffn_turn_result;

int ffn_canvas_segment_turn(ffn_canvas* canvas, const ffn_turn_request* request,
                            const int32_t* candidates /* [num_candidates][3] */,
                            ffn_turn_result* result, int32_t max_overlaps,
                            int32_t* overlap_ids, int64_t* overlap_counts,
                            int32_t* cand_flags, float* cand_seed,
                            int32_t* cand_seg);

/* Box transfers between the canvas and host arrays ([hi-lo] C-order):
 * checkpoint/restore and final save (inference.py:728-821, runner.py:433-482). */
int ffn_canvas_read_seed(ffn_canvas* canvas, const int32_t lo[3],
                         const int32_t hi[3], float* dst);
int ffn_canvas_read_segmentation(ffn_canvas* canvas, const int32_t lo[3],
                                 const int32_t hi[3], int32_t* dst);
int ffn_canvas_write_seed(ffn_canvas* canvas, const int32_t lo[3],
                          const int32_t hi[3], const float* src);
int ffn_canvas_write_segmentation(ffn_canvas* canvas, const int32_t lo[3],
                                  const int32_t hi[3], const int32_t* src);

/* ---- measurement ----------------------------------------------------------
 * HIP-event timing of the dominant kernel (the 32->32 3x3x3 MFMA conv) on the
 * engine's own stream.  mode 0 = off, 1 = one event pair per conv launch,
 * 2 = one event pair around the whole chain of 2*depth-1 conv launches of a
 * step (duration / launches then includes the inter-kernel gaps but not the
 * per-event barrier overhead of mode 1). */
int ffn_engine_set_profiling(ffn_engine* engine, int mode);
int ffn_engine_get_profile(ffn_engine* engine, double* conv_ms_total,
                           int64_t* conv_launches, int reset);
/* The individual event-pair durations (ms) behind ffn_engine_get_profile since
 * its last reset -- one per sampled conv launch (mode 1) or per sampled conv
 * chain of a step (mode 2: 2 depth - 1 launches, or the ONE launch of the
 * resident stack): *n = how many exist, the first min(*n, max_n) are copied. */
int ffn_engine_get_profile_samples(ffn_engine* engine, float* out_ms, int max_n, int* n);
/* A model whose prediction is smaller than the seed it reads (ModelInfo
 * pred_mask_size < input_seed_size; reference ffn/training/model.py:168-183 pads
 * the update with zeros around the centre, ffn/inference/inference.py:218,410-411
 * writes only the centred box): canvas steps (ffn_canvas_step*, segment_at /
 * segment_many) then score the move faces around the CENTRE OF THAT BOX, count
 * logits >= move_threshold and apply the disco bias inside it, and paste only it;
 * ffn_predict still returns the whole FoV (the caller crops).  pred_zyx = the
 * FoV: back to the default.  Deltas must fit the box's half size. */
int ffn_engine_set_pred_size(ffn_engine* engine, const int32_t pred_zyx[3]);
/* Kernel choice and tuning switches.  "conv_variant":
 *   0 = simple exact-f32 MFMA conv over padded positions (takes any FoV that
 *       fits the LDS at all),
 *   2 = exact-f32 MFMA conv over compact chunks, pipelined, K-split middle tile
 *       (bitwise the oracle's fmaf chain),
 *   6 = conv32d: every f32 product carried as 3 fp16 products (hi + 2^-11 *
 *       residual: 22 mantissa bits, about the rounding noise of an f32 GEMM) on
 *       32x32x16 MFMAs, the 27 taps split over 4 waves, producer-split fp16
 *       planes staged by LDS-DMA; operands must stay inside the fp16 range,
 *       else FFN_ERR_RANGE,
 *   7 = 6 in 96-voxel chunks (two workgroups per CU; the same bits),
 *   8 = conv32m: the same products, M split over the waves, weights through an
 *       LDS-DMA ring, two workgroups per CU,
 *   9 (default where the FoV has 257 .. 512 chunks of 128 voxels, e.g. 33^3) =
 *       a step of ONE FoV runs conv32mt (conv32m for the first 256 chunks +
 *       32-voxel K-split tail workgroups), a step of several runs conv32m,
 *  10 = a step of ONE FoV on 80-voxel workgroups, two per CU (conv32hs / conv32h:
 *       four 16-position tiles on 16x16x32 MFMAs + a fifth tile split over the
 *       waves by tap; another summation order than 9, the same tolerance); a step
 *       of several FoVs runs conv32m as under 9.  Measured SLOWER than 9 at 33^3
 *       (profiles/r06_gate_two_chains.txt): selectable, not a default,
 *  -1 = this FoV's exact-f32 kernel ("exact_variant": 2 where it fits, else 0).
 * (1, 3, 4, 5 -- conv32p, the bf16x3 and the earlier fp16 kernels -- were removed
 * in ABI 7; they are in the history up to commit af82310.)
 * "tail_batched" 1 = conv32mt for every step (one arithmetic whatever the
 * batch); "batch_chunks" (variant 6): the form batched steps take.  Results
 * are identical up to f32 summation order.  "fuse_head" (variant 2): 1x1x1
 * head inside the last conv launch.  "store_policy" (variant 2): 0 write-back,
 * 1 write-through, 2 non-temporal conv stores.
 * "speculate" (default 1): single-FoV steps made by ffn_canvas_segment_at queue
 * conv0_a of the NEXT step behind their paste, for the positions the segment
 * loop expects to pop next (the kernel takes the first that passes
 * Canvas.is_valid_pos on the pasted canvas; the host makes the same choice from
 * the step result and then queues the rest of the step behind it; a step whose
 * launch chose otherwise pastes nothing and is made again).  "fuse_paste"
 * (default 1): faces and paste of a single-FoV step as one launch; "fuse_conv0a"
 * (default 1): ... and the next step's conv0_a in it as well.  "stack_ahead"
 * (default 1): the resident conv stack of that next step is queued right behind the
 * launch that holds its conv0_a, before the host has seen this step's record -- the
 * host's turn-around and the launch latency leave the step's critical path (the stack
 * reads only what that conv0_a wrote; if the device found no valid position it ends
 * after its first conv and the step is made the ordinary way).  "paste_blocks" (0 =
 * automatic: one block per compute unit in the fused step launch).  None changes any
 * result.  "debug_submit_delay_ns": the host idles this long in front of every step's
 * launches (an experiment: what a slower host costs with and without stack_ahead);
 * "debug_fused_twice": the fused step launch is made twice (idempotent), so that a trace of
 * the second shows what warm caches are worth (nothing: profiles/r06_turn_around.txt).
 * "flow": how the 2 depth - 1 convs of a single-FoV step of conv_variant 9 run:
 *   0 = one dependent launch per conv, 1 = the same launches with the flagged
 *   hand-off compiled in, 2 = ONE resident launch whose workgroups hand rows to
 *   each other (conv32ps) -- bit-identical in every mode.  2 is the default where
 *   the device can hold all of the launch's workgroups at once (checked at
 *   ffn_engine_create: compute units x occupancy >= grid; else 0, and setting 2
 *   is FFN_ERR_ARG).  A resident launch that times out waiting for one of its
 *   own workgroups voids its step with FFN_ERR_FLOW (see above); after three such
 *   steps in a row the engine sets "flow" = 0 itself ("flow_auto_off" reads 1;
 *   setting "flow" again re-arms).  "flow_debug" 2048: fault injection for
 *   tests (a producer stops publishing); its other bits and "debug_clock" 4 act
 *   only in -DFFN_EXPERIMENTS=1 builds.
 * "flow_pace": the beat of the resident launch, in 10-ns ticks: conv l of a
 *   workgroup does not start before t0 + l x beat + "flow_pace_spread" x (its first
 *   voxel / V).  -1 (default) = the beat ffn_engine_set_weights MEASURED for this
 *   device (a ladder of beats on noise inputs, ~80 ms; 0 if none beats the
 *   free-running launch by 1.5 %), 0 = free-running, > 0 = that beat;
 *   "flow_pace_spread" -1 (default) = as wide as the beat.  Timing only: results are
 *   bit-identical under any beat (profiles/r06_pacing.txt).
 * "debug_fused_trace" N: the N-th next single-FoV step stamps when each role of its
 *   two launches ran ("debug_fused_stamp_4" .. "_27", 10-ns ticks after the stack's
 *   first workgroup -- under stack_ahead after the entry of the step's faces block:
 *   tools/gpu_step_trace.py names the slots). */
int ffn_engine_set_option(ffn_engine* engine, const char* name, int value);
/* Current value of an option ("conv_variant", "exact_variant", "fuse_head",
 * "store_policy", "sync_mode", "profile_every", "speculate", "fuse_paste",
 * "fuse_conv0a", "stack_ahead", "paste_blocks", "flow", "flow_auto_off", "flow_pace", "flow_pace_now" (the beat in
 * use), "flow_pace_free_ns" / "flow_pace_best_ns" (what the measurement saw per
 * stack: free-running, at its best beat)), "stat_flow_timeouts" (polls of the
 * resident launch that gave up, ever) / "stat_flow_voids" (steps voided by one),
 * "stat_turn_gpu_ns" / "stat_turn_host_ns" / "stat_launch_host_ns" / "stat_turn_count"
 * (between two single-FoV steps inside a segment, mean since "stat_reset": record
 * published -> first instruction of the next resident launch as the GPU saw it;
 * record seen -> that launch queued, and the launch call alone, on the host), or
 * a statistic of the step calls since set_option("stat_reset", 0):
 * "stat_step_calls", "stat_step_items" (FoVs in them), "stat_hist_<n>" (calls
 * with n FoVs), "stat_spec_launched" / "stat_spec_hits" / "stat_spec_mismatch"
 * (conv0_a launches made ahead, steps that ran on one, steps repeated),
 * "stat_ahead_used" / "stat_ahead_wasted" (stacks queued ahead that their step used /
 * that no step used). */
int ffn_engine_get_option(ffn_engine* engine, const char* name, int* value);
/* Debug: with option "debug_clock" = 1 the compact conv kernel records, for its
 * first workgroup, per wave {shader clock at entry, at main-loop start, at
 * main-loop end, at exit, wall clock (100 MHz) at entry, at exit}. */
int ffn_engine_debug_clocks(ffn_engine* engine, long long* out24);
/* Debug: with "debug_clock" = 2 every workgroup of the conv32m / conv32mt launch
 * of layer "debug_layer" also records {wall clock (100 MHz) at entry, at exit,
 * HW_ID, XCC_ID}: out[4 b ..] for blockIdx b < max_wgs <= 4096 (zeros for a
 * block that exited at once).  Clears the records. */
int ffn_engine_debug_workgroups(ffn_engine* engine, long long* out, int max_wgs);
/* Debug: with "debug_clock" = 4 every workgroup of a FLOW conv (option "flow" 1
 * or 2: the resident stack of a single-FoV step) records per conv of the stack
 * six wall-clock stamps (100 MHz): body entry, input tiles seen, first segment
 * landed, tap loop over, stores drained, tiles published.
 * out[(slot * 64 + conv) * 8 + 0..5] for workgroup slot < max_slots (main chunks
 * first, then the tail chunks).  Clears the records. */
int ffn_engine_debug_flow_trace(ffn_engine* engine, long long* out, int max_slots);
/* Blocks until all work queued on the engine's stream has finished. */
int ffn_engine_synchronize(ffn_engine* engine);

/* Thread-local description of the last error. */
const char* ffn_last_error(void);
/* ABI version of this header. */
int ffn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FFN_HIP_H_ */
