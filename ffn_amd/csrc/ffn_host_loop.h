// Canvas.segment_at on the host side of the library: the FoV loop of one
// segment without an interpreter between the steps.
//
// What runs here is the reference's inner loop (ffn/inference/inference.py:
// 460-533) with its default movement policy (FaceMaxMovementPolicy,
// ffn/inference/movement.py:166-222) and position test (Canvas.is_valid_pos,
// inference.py:312-346): a FIFO of (score, position) candidates, a visited set
// of positions quantised to the delta grid, six face maxima per step sorted by
// descending (score, offset) with duplicates dropped.  Every decision is the
// same integer / f32 comparison the Python mirror (ffn_amd/inference) makes, in
// the same order, so the visited positions are identical step for step.
//
// Pure C++17, no HIP: the device is a template parameter with
//   int step(const ffn_step_request&, const ffn_step_params&, ffn_step_result*)
//   int read_point(const int32_t pos[3], float* seed, int32_t* seg)
// (FFN_OK or an error code that is passed through).  libffn_hip.so instantiates
// it with the HIP canvas; tests/host_loop_shim.cpp with callbacks into the
// emulated device, so the loop itself is covered without a GPU.
#pragma once

#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/ffn_hip.h"

namespace ffn_host {

struct Coord {
  int32_t z, y, x;
  bool operator==(const Coord& o) const { return z == o.z && y == o.y && x == o.x; }
};

struct CoordHash {
  size_t operator()(const Coord& c) const {
    uint64_t h = (uint64_t)(uint32_t)c.z * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)c.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (uint64_t)(uint32_t)c.x * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return (size_t)h;
  }
};

inline int32_t floordiv(int32_t a, int32_t b) {  // Python's //, b > 0
  int32_t q = a / b;
  if ((a % b != 0) && (a < 0)) --q;
  return q;
}

// State of one segment's loop; lives in the canvas between calls so that a call
// bounded by max_steps can be resumed.
struct SegmentState {
  struct Entry {
    float score;
    Coord pos;
    Coord q;  // pos quantised to the delta grid (movement.py:200-208)
  };
  std::deque<Entry> queue;
  std::unordered_set<Coord, CoordHash> done;
  // post-step values at positions the next validity tests may ask for
  // (DeviceCanvas._cache): seed logit and segmentation id
  std::unordered_map<Coord, std::pair<float, int32_t>, CoordHash> cache;
  Coord start{0, 0, 0};
  float start_logit = 0.f;
  bool start_logit_known = false;
  bool active = false;     // a segment is in progress (resumable)
  bool has_pending = false;  // a popped position whose step did not complete
  Coord pending{0, 0, 0};
  int32_t min_pos[3] = {0, 0, 0}, max_pos[3] = {0, 0, 0};
  // history of the segment (keep_history): positions and deleted-voxel counts
  std::vector<int32_t> history;          // 3 per step
  std::vector<uint32_t> history_deleted;  // 1 per step
  // segment_many: tallies since the segment's start, over all the calls it took
  int64_t many_steps = 0, many_skip_threshold = 0, many_skip_invalid_pos = 0,
          many_gate_rejects = 0;
};

constexpr int kHintMax = 3;  // positions per Dev::hint_next call

// One prepared FoV step of a segment loop: the position popped for it, the queue
// head whose values ride along, the request as the device gets it.
struct StepPending {
  Coord pos;
  Coord cands[FFN_MAX_CANDIDATES];
  int nc = 0;
  ffn_step_request req;
};

// A batched step of segment_many that has been QUEUED but whose results have not
// been looked at: it stays in flight across the return to the caller, who does
// the between-segment work of the loop that has just ended meanwhile; the next
// segment_many call (or whoever needs one of these canvases first) waits for it
// and feeds the results to the loops it belongs to.
struct ManyCarry {
  struct Item {
    SegmentState* st;
    ffn_segment_params params;
    StepPending pd;
  };
  bool active = false;
  std::vector<Item> items;  // in the order of the batch
};

template <class Dev>
class SegmentLoop {
 public:
  SegmentLoop(Dev& dev, SegmentState& st, const ffn_segment_params& p)
      : dev_(dev), st_(st), p_(p) {
    for (int a = 0; a < 3; ++a) {
      d_[a] = p.deltas_zyx[a];
      dh_[a] = d_[a] / 2;
      dm_[a] = d_[a] > 1 ? d_[a] : 1;
    }
  }

  // Starts (resume = 0) or continues (resume = 1) the segment at `start`.
  int run(const int32_t start[3], int resume, ffn_segment_result* out) {
    int rc = begin(start, resume, out);
    if (rc) return rc;
    for (;;) {
      Pending pd;
      bool ended = false;
      rc = prepare(&pd, out, &ended);
      if (rc || ended) break;
      // what the step AFTER this one will most likely be made at: the device may
      // start on it before this step's result has come back (HipLoopDevice)
      int32_t hint[kHintMax][3];
      const int nh = guess_next(pd.pos, hint);
      dev_.hint_next(nh, hint);
      ffn_step_result res;
      rc = dev_.step(pd.req, p_.step, &res);
      if (rc) {  // nothing was pasted: the position stays pending
        step_failed(pd);
        break;
      }
      consume(pd, res);
    }
    finish(out);
    return rc;
  }

  // ---- the same loop in phases, for a caller that batches the steps of several
  // canvases into one engine call (ffn_canvas_segment_many) -----------------------
  typedef StepPending Pending;

  int begin(const int32_t start[3], int resume, ffn_segment_result* out) {
    std::memset(out, 0, sizeof(*out));
    steps_ = 0;
    if (!resume) {
      st_.queue.clear();
      st_.done.clear();
      st_.cache.clear();
      st_.history.clear();
      st_.history_deleted.clear();
      st_.start = Coord{start[0], start[1], start[2]};
      st_.has_pending = false;
      st_.active = true;
      for (int a = 0; a < 3; ++a) {
        st_.min_pos[a] = p_.init_min_pos[a];
        st_.max_pos[a] = p_.init_max_pos[a];
      }
      st_.start_logit_known = p_.initial_start_logit == p_.initial_start_logit;
      st_.start_logit = p_.initial_start_logit;
      // the first move: the start position itself, at twice the threshold
      // (inference.py:481-483)
      push((float)(p_.score_threshold * 2.0), st_.start);
    } else if (!st_.active) {
      return FFN_ERR_STATE;
    }
    return FFN_OK;
  }

  // The next FoV step's request, or *ended (queue empty / seed too weak / step
  // budget spent).  A device read that fails keeps the popped position pending.
  int prepare(Pending* pd, ffn_segment_result* out, bool* ended) {
    *ended = false;
    if (p_.max_steps > 0 && steps_ >= p_.max_steps) {
      out->budget_exhausted = 1;
      *ended = true;
      return FFN_OK;
    }
    Coord pos;
    if (st_.has_pending) {
      pos = st_.pending;
    } else {
      bool found = false;
      const int rc = next(&pos, &found, out);
      if (rc) return rc;
      if (!found) {
        st_.active = false;
        *ended = true;
        return FFN_OK;
      }
    }
    // "seed got too weak" (inference.py:503-505)
    if (!st_.start_logit_known) {
      int32_t seg;
      const int32_t sp[3] = {st_.start.z, st_.start.y, st_.start.x};
      const int rc = dev_.read_point(sp, &st_.start_logit, &seg);
      if (rc) {
        st_.pending = pos;
        st_.has_pending = true;
        return rc;
      }
      st_.start_logit_known = true;
    }
    if (st_.start_logit < p_.step.move_threshold) {
      out->seed_got_too_weak = 1;
      st_.has_pending = false;
      st_.active = false;
      *ended = true;
      return FFN_OK;
    }
    // ---- one FoV step ----------------------------------------------------------
    pd->pos = pos;
    ffn_step_request& req = pd->req;
    req.pos[0] = pos.z, req.pos[1] = pos.y, req.pos[2] = pos.x;
    req.start_pos[0] = st_.start.z, req.start_pos[1] = st_.start.y,
    req.start_pos[2] = st_.start.x;
    pd->nc = peek(pd->cands, p_.prefetch < FFN_MAX_CANDIDATES
                                 ? p_.prefetch : FFN_MAX_CANDIDATES);
    req.num_candidates = pd->nc;
    for (int k = 0; k < pd->nc; ++k) {
      req.candidates[k][0] = pd->cands[k].z;
      req.candidates[k][1] = pd->cands[k].y;
      req.candidates[k][2] = pd->cands[k].x;
    }
    // until the step has been made the position counts as pending: a caller
    // that returns between prepare and consume loses nothing
    st_.pending = pos;
    st_.has_pending = true;
    return FFN_OK;
  }

  void step_failed(const Pending& pd) {
    st_.pending = pd.pos;
    st_.has_pending = true;
  }

  void consume(const Pending& pd, const ffn_step_result& res) {
    const Coord& pos = pd.pos;
    st_.has_pending = false;
    ++steps_;
    st_.cache.clear();
    for (int k = 0; k < pd.nc; ++k)
      st_.cache.emplace(pd.cands[k],
                        std::make_pair(res.cand_seed[k], res.cand_seg[k]));
    st_.start_logit = res.start_logit;
    st_.start_logit_known = true;
    const int32_t pv[3] = {pos.z, pos.y, pos.x};
    for (int a = 0; a < 3; ++a) {
      if (pv[a] < st_.min_pos[a]) st_.min_pos[a] = pv[a];
      if (pv[a] > st_.max_pos[a]) st_.max_pos[a] = pv[a];
    }
    if (p_.keep_history) {
      st_.history.insert(st_.history.end(), pv, pv + 3);
      st_.history_deleted.push_back(res.num_deleted);
    }
    update(res, pos);
  }

  void finish(ffn_segment_result* out) const {
    out->num_steps = steps_;
    out->start_logit = st_.start_logit;
    out->start_logit_known = st_.start_logit_known ? 1 : 0;
    out->active = st_.active ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
      out->min_pos[a] = st_.min_pos[a];
      out->max_pos[a] = st_.max_pos[a];
    }
    out->queue_len = (int64_t)st_.queue.size();
  }

 private:
  Coord quantize(const Coord& c) const {
    return Coord{floordiv(c.z - st_.start.z + dh_[0], dm_[0]),
                 floordiv(c.y - st_.start.y + dh_[1], dm_[1]),
                 floordiv(c.x - st_.start.x + dh_[2], dm_[2])};
  }
  void push(float score, const Coord& c) {
    st_.queue.push_back(SegmentState::Entry{score, c, quantize(c)});
  }

  int read_cached(const Coord& c, float* seed, int32_t* seg) {
    auto it = st_.cache.find(c);
    if (it != st_.cache.end()) {
      *seed = it->second.first;
      *seg = it->second.second;
      return FFN_OK;
    }
    const int32_t pv[3] = {c.z, c.y, c.x};
    const int rc = dev_.read_point(pv, seed, seg);
    if (rc) return rc;
    st_.cache.emplace(c, std::make_pair(*seed, *seg));
    return FFN_OK;
  }

  bool in_bounds(const Coord& c) const {
    const int32_t pv[3] = {c.z, c.y, c.x};
    for (int a = 0; a < 3; ++a)
      if (pv[a] - p_.margin_zyx[a] < 0 || pv[a] + p_.margin_zyx[a] >= p_.shape_zyx[a])
        return false;
    return true;
  }

  // Canvas.is_valid_pos (inference.py:312-346), device-state part included
  int is_valid(const Coord& c, bool* ok, ffn_segment_result* out) {
    *ok = false;
    float seed;
    int32_t seg;
    int rc = read_cached(c, &seed, &seg);
    if (rc) return rc;
    if (seed < p_.step.move_threshold) {
      ++out->skip_threshold;
      if (in_bounds(c)) ++out->gate_rejects;
      return FFN_OK;
    }
    if (!in_bounds(c)) {
      ++out->skip_invalid_pos;
      return FFN_OK;
    }
    if (seg > 0) {
      ++out->skip_invalid_pos;
      ++out->gate_rejects;
      return FFN_OK;
    }
    *ok = true;
    return FFN_OK;
  }

  // FaceMaxMovementPolicy.__next__ (movement.py:182-198)
  int next(Coord* pos, bool* found, ffn_segment_result* out) {
    *found = false;
    while (!st_.queue.empty()) {
      const SegmentState::Entry e = st_.queue.front();
      if (st_.done.count(e.q)) {
        st_.queue.pop_front();
        continue;
      }
      bool ok = false;
      // the entry leaves the queue only once its test could be made
      const int rc = is_valid(e.pos, &ok, out);
      if (rc) return rc;
      st_.queue.pop_front();
      if (ok) {
        *pos = e.pos;
        *found = true;
        return FFN_OK;
      }
    }
    return FFN_OK;
  }

  // The first kHintMax queued positions that next() can still accept once the
  // step at `cur` has been made: not visited (`cur` counts as visited by then),
  // FoV inside the canvas.  Which of them is valid depends on the step's result;
  // next() takes the first that is, moves queued by the step come after them.
  // ... and not KNOWN to stay invalid: the values the last step's record brought for the
  // head of the queue (and the moves it queued) still hold where the step at `cur` cannot
  // write -- a segmentation id > 0 stays (ids change at a commit), a seed below the move
  // threshold stays unless the position lies inside the FoV about to be pasted.  Without
  // this, 4.7 % of the steps found all three hinted positions invalid while the queue held
  // a valid one further down (bench.py: speculation.launched_but_step_elsewhere_hint_list_full).
  int guess_next(const Coord& cur, int32_t (*out)[3]) const {
    const Coord qc = quantize(cur);
    int n = 0;
    for (const auto& e : st_.queue) {
      if (n >= kHintMax) break;
      if (e.q == qc || st_.done.count(e.q) || !in_bounds(e.pos)) continue;
      const auto it = st_.cache.find(e.pos);
      if (it != st_.cache.end()) {
        const float seed = it->second.first;
        const int32_t seg = it->second.second;
        if (seg > 0) continue;
        const bool in_fov = std::abs(e.pos.z - cur.z) <= p_.margin_zyx[0] &&
                            std::abs(e.pos.y - cur.y) <= p_.margin_zyx[1] &&
                            std::abs(e.pos.x - cur.x) <= p_.margin_zyx[2];
        if (seed < p_.step.move_threshold && !in_fov) continue;
      }
      out[n][0] = e.pos.z, out[n][1] = e.pos.y, out[n][2] = e.pos.x;
      ++n;
    }
    return n;
  }

  // first `limit` queued positions not yet visited (their post-step seed /
  // segmentation values come back with the step result)
  int peek(Coord* out, int limit) {
    while (!st_.queue.empty() && st_.done.count(st_.queue.front().q))
      st_.queue.pop_front();
    int n = 0;
    for (const auto& e : st_.queue) {
      if (n >= limit) break;
      if (st_.done.count(e.q)) continue;
      out[n++] = e.pos;
    }
    return n;
  }

  // FaceMaxMovementPolicy.update (movement.py:210-222) on the six face maxima
  // of get_scored_move_offsets (movement.py:42-100)
  void update(const ffn_step_result& res, const Coord& pos) {
    st_.done.insert(quantize(pos));
    struct Move {
      float score;
      int32_t rel[3];
      int32_t seg;
    };
    Move moves[6];
    int nm = 0;
    int k = 0;
    for (int axis = 0; axis < 3; ++axis) {
      const int o0 = axis == 0 ? 1 : 0;
      const int o1 = axis == 2 ? 1 : 2;
      for (int sign = -1; sign <= 1; sign += 2, ++k) {
        const int32_t off = sign * d_[axis];
        if (off == 0) continue;
        const float score = res.face_score[k];
        if ((double)score < p_.score_threshold) continue;
        const int32_t ncols = 2 * d_[o1] + 1;
        const int32_t fi = res.face_index[k] / ncols;
        const int32_t fj = res.face_index[k] - fi * ncols;
        Move m;
        m.score = score;
        m.rel[axis] = off;
        m.rel[o0] = fi - d_[o0];
        m.rel[o1] = fj - d_[o1];
        m.seg = res.face_seg[k];
        moves[nm++] = m;
      }
    }
    // descending (score, offset, seg), as sorted(..., reverse=True) of tuples
    std::sort(moves, moves + nm, [](const Move& a, const Move& b) {
      if (a.score != b.score) return a.score > b.score;
      for (int i = 0; i < 3; ++i)
        if (a.rel[i] != b.rel[i]) return a.rel[i] > b.rel[i];
      return a.seg > b.seg;
    });
    for (int i = 0; i < nm; ++i) {
      // one voxel shared by two faces: keep the first (movement.py:98-100)
      if (i > 0 && moves[i].score == moves[i - 1].score &&
          moves[i].rel[0] == moves[i - 1].rel[0] &&
          moves[i].rel[1] == moves[i - 1].rel[1] &&
          moves[i].rel[2] == moves[i - 1].rel[2])
        continue;
      const Coord c{moves[i].rel[0] + pos.z, moves[i].rel[1] + pos.y,
                    moves[i].rel[2] + pos.x};
      push(moves[i].score, c);
      // freshly queued: its seed logit is the face maximum just pasted
      st_.cache.emplace(c, std::make_pair(moves[i].score, moves[i].seg));
    }
  }

  Dev& dev_;
  SegmentState& st_;
  const ffn_segment_params& p_;
  int32_t d_[3], dh_[3], dm_[3];
  int64_t steps_ = 0;
};

// The segment loops of n canvases advanced TOGETHER (ffn_canvas_segment_many;
// config C3: the reference's client threads + batching server thread,
// executor.py:266-340, with the per-canvas queues in this library): every
// round asks each running loop for its next FoV step and makes ONE batched
// step for them.
//   batch_step(nb, idx, reqs, step_params, results) -> rc steps canvases
//   idx[0 .. nb) with reqs[0 .. nb).
// Returns as soon as at least one loop has ended -- queue empty, seed too weak,
// or its max_steps spent in this call -- so that the caller can finish that
// segment and hand the canvas its next one; finished[k] says which.  The others
// stay resumable (resume[k] = 1 on the next call, or later: a canvas may sit
// out any number of calls).  out[k] always counts from the START of canvas k's
// segment (all its calls), budget_exhausted aside.  Every canvas must use the
// same step parameters (one engine call = one set); an error (e.g.
// FFN_ERR_RANGE) leaves every prepared position pending: after the caller has
// dealt with it the same call with resume = 1 everywhere repeats the step.
// The results of a carried step to the loops it was made for (whichever call they
// are part of next): wait(results) -> rc for carry->items.size() records.  An
// error means nothing was pasted: the positions stay pending, as after a failed
// batch_step.
template <class Dev, class Wait>
int resolve_carry(ManyCarry* carry, Dev& any_dev, Wait&& wait) {
  if (!carry || !carry->active) return FFN_OK;
  std::vector<ffn_step_result> res(carry->items.size());
  const int rc = wait(res.data());
  carry->active = false;
  if (rc == FFN_OK) {
    for (size_t b = 0; b < carry->items.size(); ++b) {
      ManyCarry::Item& it = carry->items[b];
      SegmentLoop<Dev> loop(any_dev, *it.st, it.params);  // (consume reads no device)
      loop.consume(it.pd, res[b]);
      it.st->many_steps += 1;
    }
  }
  carry->items.clear();
  return rc;
}

// carry != NULL: when a loop ends, the step the others have prepared is QUEUED
// (submit(nb, idx, reqs, step_params) -> rc) and the call returns without waiting
// for it; the next call starts with resolve_carry.  batch_step = submit + wait.
template <class Dev, class BatchStep, class Submit, class Wait>
int segment_many(int n, Dev* devs, SegmentState* const* states,
                 const int32_t (*starts)[3], const ffn_segment_params* params,
                 const int32_t* resume, ffn_segment_result* out, int32_t* finished,
                 BatchStep&& batch_step, ManyCarry* carry, Submit&& submit,
                 Wait&& wait) {
  typedef SegmentLoop<Dev> Loop;
  for (int k = 1; k < n; ++k)
    if (std::memcmp(&params[k].step, &params[0].step, sizeof(ffn_step_params)) != 0)
      return FFN_ERR_ARG;
  const int rc_carry = resolve_carry(carry, devs[0], wait);
  std::vector<Loop> loops;
  loops.reserve(n);
  std::vector<char> running(n, 1);
  std::vector<typename Loop::Pending> pend(n);
  std::vector<int> idx(n);
  std::vector<ffn_step_request> reqs(n);
  std::vector<ffn_step_result> results(n);
  int rc = FFN_OK;
  for (int k = 0; k < n; ++k) {
    finished[k] = 0;
    loops.emplace_back(devs[k], *states[k], params[k]);
    rc = loops[k].begin(starts[k], resume[k], &out[k]);
    if (rc) return rc;
    if (!resume[k])
      states[k]->many_steps = states[k]->many_skip_threshold =
          states[k]->many_skip_invalid_pos = states[k]->many_gate_rejects = 0;
  }
  bool any_ended = false;
  rc = rc_carry;  // (a voided carried step: no step is made in this call)
  while (!any_ended && rc == FFN_OK) {
    int nb = 0;
    for (int k = 0; k < n && rc == FFN_OK; ++k) {
      if (!running[k]) continue;
      bool ended = false;
      rc = loops[k].prepare(&pend[k], &out[k], &ended);
      if (rc) break;
      if (ended) {
        running[k] = 0;
        finished[k] = 1;
        any_ended = true;
        continue;
      }
      idx[nb] = k;
      reqs[nb] = pend[k].req;
      ++nb;
    }
    if (rc || nb == 0) break;
    // (the canvases that did prepare a step make it even when another loop has
    // just ended: their positions are popped, the batch slot costs nothing)
    if (carry && any_ended) {
      rc = submit(nb, idx.data(), reqs.data(), params[0].step);
      if (rc) break;
      carry->items.clear();
      for (int b = 0; b < nb; ++b)
        carry->items.push_back(
            ManyCarry::Item{states[idx[b]], params[idx[b]], pend[idx[b]]});
      carry->active = true;
      break;
    }
    rc = batch_step(nb, idx.data(), reqs.data(), params[0].step, results.data());
    if (rc) break;  // nothing was pasted: every prepared position stays pending
    for (int b = 0; b < nb; ++b) loops[idx[b]].consume(pend[idx[b]], results[b]);
  }
  for (int k = 0; k < n; ++k) {
    loops[k].finish(&out[k]);
    SegmentState& st = *states[k];
    st.many_steps += out[k].num_steps;
    st.many_skip_threshold += out[k].skip_threshold;
    st.many_skip_invalid_pos += out[k].skip_invalid_pos;
    st.many_gate_rejects += out[k].gate_rejects;
    out[k].num_steps = st.many_steps;
    out[k].skip_threshold = st.many_skip_threshold;
    out[k].skip_invalid_pos = st.many_skip_invalid_pos;
    out[k].gate_rejects = st.many_gate_rejects;
  }
  return rc;
}

template <class Dev, class BatchStep>
int segment_many(int n, Dev* devs, SegmentState* const* states,
                 const int32_t (*starts)[3], const ffn_segment_params* params,
                 const int32_t* resume, ffn_segment_result* out, int32_t* finished,
                 BatchStep&& batch_step) {
  auto no_submit = [](int, const int*, const ffn_step_request*,
                      const ffn_step_params&) { return (int)FFN_ERR_STATE; };
  auto no_wait = [](ffn_step_result*) { return (int)FFN_ERR_STATE; };
  return segment_many(n, devs, states, starts, params, resume, out, finished,
                      batch_step, static_cast<ManyCarry*>(nullptr), no_submit, no_wait);
}

}  // namespace ffn_host
