// TEST INFRASTRUCTURE: ffn_amd/csrc/ffn_host_loop.h instantiated over callbacks,
// so that the library's segment_at loop -- the same template libffn_hip.so
// instantiates over the HIP canvas -- runs against the emulated device of
// tests/emulated_device.py without a GPU.  Built by tests/test_host_loop.py
// with g++; never part of the product.
#include "../ffn_amd/csrc/ffn_host_loop.h"

extern "C" {

typedef int (*shim_step_fn)(const ffn_step_request*, const ffn_step_params*,
                            ffn_step_result*);
typedef int (*shim_read_fn)(const int32_t*, float*, int32_t*);

// (optional) what the loop expects to pop after the step it is about to make
typedef void (*shim_hint_fn)(int, const int32_t*);
static shim_hint_fn g_hint_cb = nullptr;
void shim_set_hint_cb(shim_hint_fn fn) { g_hint_cb = fn; }

struct ShimDevice {
  shim_step_fn step_cb;
  shim_read_fn read_cb;
  void hint_next(int n, const int32_t (*pos)[3]) {
    if (g_hint_cb) g_hint_cb(n, &pos[0][0]);
  }
  int step(const ffn_step_request& req, const ffn_step_params& params,
           ffn_step_result* res) {
    return step_cb(&req, &params, res);
  }
  int read_point(const int32_t pos[3], float* seed, int32_t* seg) {
    return read_cb(pos, seed, seg);
  }
};

void* shim_state_create() { return new ffn_host::SegmentState(); }
void shim_state_destroy(void* s) { delete static_cast<ffn_host::SegmentState*>(s); }

int shim_segment_at(void* state, shim_step_fn step_cb, shim_read_fn read_cb,
                    const int32_t start[3], const ffn_segment_params* p,
                    int resume, ffn_segment_result* out) {
  ShimDevice dev{step_cb, read_cb};
  auto& st = *static_cast<ffn_host::SegmentState*>(state);
  ffn_host::SegmentLoop<ShimDevice> loop(dev, st, *p);
  return loop.run(start, resume, out);
}

// ffn_host::segment_many (what ffn_canvas_segment_many runs over HIP canvases)
// over callbacks: ONE batched-step callback, point reads by canvas index.
typedef int (*shim_batch_fn)(int, const int*, const ffn_step_request*,
                             const ffn_step_params*, ffn_step_result*);
typedef int (*shim_read_k_fn)(int, const int32_t*, float*, int32_t*);

struct ShimManyDevice {
  shim_read_k_fn read_cb;
  int k;
  int read_point(const int32_t pos[3], float* seed, int32_t* seg) {
    return read_cb(k, pos, seed, seg);
  }
};

int shim_segment_many(int n, void* const* states, shim_batch_fn batch_cb,
                      shim_read_k_fn read_cb, const int32_t (*starts)[3],
                      const ffn_segment_params* params, const int32_t* resume,
                      ffn_segment_result* out, int32_t* finished) {
  std::vector<ShimManyDevice> devs(n);
  std::vector<ffn_host::SegmentState*> st(n);
  for (int k = 0; k < n; ++k) {
    devs[k] = ShimManyDevice{read_cb, k};
    st[k] = static_cast<ffn_host::SegmentState*>(states[k]);
  }
  auto batch_step = [&](int nb, const int* idx, const ffn_step_request* reqs,
                        const ffn_step_params& sp, ffn_step_result* res) {
    return batch_cb(nb, idx, reqs, &sp, res);
  };
  return ffn_host::segment_many(n, devs.data(), st.data(), starts, params, resume, out,
                                finished, batch_step);
}

// ... with the step the running loops have prepared left "in flight" when a loop
// ends (ffn_canvas_segment_many_carry).  The emulated device makes the step when
// it is queued; its results (or, with defer_error, its error) are handed over
// when the next call waits for it -- the protocol of the real thing, whose
// between-segment work touches another canvas than the step in flight.
static ffn_host::ManyCarry g_carry;
static std::vector<ffn_step_result> g_carry_res;
static int g_carry_rc = 0;

int shim_carry_active() { return g_carry.active ? 1 : 0; }

int shim_segment_many_carry(int n, void* const* states, shim_batch_fn batch_cb,
                            shim_read_k_fn read_cb, const int32_t (*starts)[3],
                            const ffn_segment_params* params, const int32_t* resume,
                            ffn_segment_result* out, int32_t* finished,
                            int defer_error) {
  std::vector<ShimManyDevice> devs(n);
  std::vector<ffn_host::SegmentState*> st(n);
  for (int k = 0; k < n; ++k) {
    devs[k] = ShimManyDevice{read_cb, k};
    st[k] = static_cast<ffn_host::SegmentState*>(states[k]);
  }
  auto batch_step = [&](int nb, const int* idx, const ffn_step_request* reqs,
                        const ffn_step_params& sp, ffn_step_result* res) {
    return batch_cb(nb, idx, reqs, &sp, res);
  };
  auto submit = [&](int nb, const int* idx, const ffn_step_request* reqs,
                    const ffn_step_params& sp) {
    g_carry_res.assign(nb, ffn_step_result());
    g_carry_rc = batch_cb(nb, idx, reqs, &sp, g_carry_res.data());
    if (g_carry_rc && !defer_error) return g_carry_rc;
    return 0;
  };
  auto wait = [&](ffn_step_result* res) {
    if (g_carry_rc == 0)
      std::memcpy(res, g_carry_res.data(), sizeof(ffn_step_result) * g_carry_res.size());
    return g_carry_rc;
  };
  return ffn_host::segment_many(n, devs.data(), st.data(), starts, params, resume, out,
                                finished, batch_step, &g_carry, submit, wait);
}

size_t shim_history(void* state, int32_t* pos, uint32_t* deleted, size_t cap) {
  auto& st = *static_cast<ffn_host::SegmentState*>(state);
  const size_t n = st.history_deleted.size() < cap ? st.history_deleted.size() : cap;
  if (n) {
    std::memcpy(pos, st.history.data(), 12 * n);
    std::memcpy(deleted, st.history_deleted.data(), 4 * n);
  }
  return st.history_deleted.size();
}

}  // extern "C"

extern "C" size_t shim_sizeof_params() { return sizeof(ffn_segment_params); }
extern "C" size_t shim_sizeof_result() { return sizeof(ffn_segment_result); }
