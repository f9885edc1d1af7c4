/*
 * TEST INFRASTRUCTURE -- a stand-in for libffn_hip.so on a machine without a GPU.
 *
 * The four entry points INTEGRATION.md section A binds (include/ffn_hip.h:
 * ffn_engine_create, ffn_engine_set_weights, ffn_predict, ffn_last_error, plus
 * ffn_engine_destroy), with the header's argument meaning and error behaviour, the
 * arithmetic supplied by the ORACLE's C conv stack (oracle/convstack_oracle.c is
 * compiled into this library by tests/test_integration_snippet.py).  It lets the
 * reference's own ThreadingBatchExecutor / Canvas run behind the snippet exactly as
 * printed, so that the snippet's argtypes, shapes and locking are exercised on every
 * CPU run.  Never shipped, never loaded by ffn_amd/.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int ffn_oracle_forward(const float *image, const float *seed, int n, int Z, int Y, int X,
                       int depth, int features, const float *weights, float *logits_out,
                       int stop_after, float *act_out);
size_t ffn_oracle_weight_count(int depth, int features);

typedef struct stub_engine {
  int fov[3], depth, features, max_batch;
  float *weights;
} stub_engine;

static char g_err[256] = "";
static int g_calls = 0;

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

const char *ffn_last_error(void) { return g_err; }
int ffn_stub_predict_calls(void) { return g_calls; }

int ffn_engine_create(int device_id, const int32_t fov_zyx[3], const int32_t deltas_zyx[3],
                      int depth, int features, int max_batch, stub_engine **out) {
  (void)device_id;
  if (!fov_zyx || !deltas_zyx || !out) return fail(-1, "null argument");
  if (depth < 1 || features != 32 || max_batch < 1) return fail(-1, "bad depth / features / batch");
  for (int a = 0; a < 3; ++a)
    if (fov_zyx[a] < 3 || !(fov_zyx[a] & 1) || deltas_zyx[a] < 1) return fail(-1, "bad fov / deltas");
  stub_engine *e = (stub_engine *)calloc(1, sizeof(*e));
  memcpy(e->fov, fov_zyx, sizeof(e->fov));
  e->depth = depth;
  e->features = features;
  e->max_batch = max_batch;
  *out = e;
  return 0;
}

void ffn_engine_destroy(stub_engine *e) {
  if (!e) return;
  free(e->weights);
  free(e);
}

int ffn_engine_set_weights(stub_engine *e, const float *blob, size_t count) {
  if (!e || !blob) return fail(-1, "null argument");
  if (count != ffn_oracle_weight_count(e->depth, e->features)) {
    snprintf(g_err, sizeof(g_err), "weight blob has %zu floats, expected %zu", count,
             ffn_oracle_weight_count(e->depth, e->features));
    return -1;
  }
  free(e->weights);
  e->weights = (float *)malloc(count * sizeof(float));
  memcpy(e->weights, blob, count * sizeof(float));
  return 0;
}

int ffn_predict(stub_engine *e, int n, const float *seed, const float *image, float *logits_out) {
  if (!e || !seed || !image || !logits_out) return fail(-1, "null argument");
  if (n < 1 || n > e->max_batch) return fail(-1, "batch outside [1, max_batch]");
  if (!e->weights) return fail(-3, "weights not set");
  ++g_calls;
  return ffn_oracle_forward(image, seed, n, e->fov[0], e->fov[1], e->fov[2], e->depth,
                            e->features, e->weights, logits_out, -1, NULL)
             ? fail(-2, "oracle forward failed")
             : 0;
}
