/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * Plain-C, f32 CPU restatement of the FFN conv-stack forward pass:
 *   reference ffn/training/models/convstack_3d.py:26-56  (_predict_object_mask)
 *   reference ffn/training/models/convstack_3d.py:83-95  (concat + update_seed)
 *   reference ffn/training/model.py:168-183              (update_seed, equal sizes)
 *
 * The arithmetic itself lives in un-vendored third-party code (tensorflow >=1.4,
 * tf-slim >=1.1: tf_slim.convolution3d = NDHWC, stride 1, padding SAME,
 * cross-correlation, + bias, ReLU unless activation_fn=None).  TensorFlow is not
 * available in this environment and the reference ships no test or golden
 * vector for this forward pass, so: PARITY UNPINNED against TensorFlow.  The
 * restatement is cross-checked against an independent implementation
 * (torch.nn.functional.conv3d in f32 and f64, tests/test_oracle.py) and uses
 * the reference's shipped FIB-25 weights.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Layouts:
 *   image, seed, logits : [n][Z][Y][X] f32
 *   activations         : [Z][Y][X][C] f32 (channels last, as TF NDHWC)
 *   weights blob        : conv0_a W[3][3][3][2][F]  b[F]
 *                         conv0_b W[3][3][3][F][F]  b[F]
 *                         (conv{i}_a W,b  conv{i}_b W,b) for i in 1..depth-1
 *                         conv_lom W[F] b[1]
 *                         -- each W exactly as TF stores it (DHWIO).
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXF 64

/* out[z][y][x][co] = act( bias[co] + sum_{dz,dy,dx,ci} in[z+dz][y+dy][x+dx][ci]
 *                                         * w[dz+1][dy+1][dx+1][ci][co] )  (+ skip)
 * relu_in applies max(0,.) to the input on the fly (tf.nn.relu before conv_a).
 * SAME padding is realised by copying the input into a zero-haloed scratch
 * (adding 0*w leaves an f32 accumulator unchanged), so the tap loops carry no
 * bounds tests; XB output voxels along x share each weight row. */
#define XB 3
static void conv3x3x3(const float *in, float *out, const float *skip,
                      const float *w, const float *bias, int Z, int Y, int X,
                      int cin, int cout, int relu_in, int relu_out) {
  const int Zp = Z + 2, Yp = Y + 2, Xp = X + 2;
  float *pad = (float *)calloc((size_t)Zp * Yp * Xp * cin, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y) {
      const float *src = in + ((size_t)z * Y + y) * X * cin;
      float *dst = pad + (((size_t)(z + 1) * Yp + (y + 1)) * Xp + 1) * cin;
      for (int i = 0; i < X * cin; ++i) {
        float a = src[i];
        dst[i] = (relu_in && a < 0.0f) ? 0.0f : a;
      }
    }
#pragma omp parallel for collapse(2) schedule(static)
  for (int z = 0; z < Z; ++z) {
    for (int y = 0; y < Y; ++y) {
      for (int x0 = 0; x0 < X; x0 += XB) {
        const int nb = (X - x0) < XB ? (X - x0) : XB;
        float acc[XB][MAXF];
        for (int b = 0; b < XB; ++b)
          for (int co = 0; co < cout; ++co) acc[b][co] = 0.0f;
        for (int kz = 0; kz < 3; ++kz)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const float *ip =
                  pad + (((size_t)(z + kz) * Yp + (y + ky)) * Xp + (x0 + kx)) * cin;
              const float *wp = w + (size_t)((kz * 3 + ky) * 3 + kx) * cin * cout;
              if (cout == 32 && nb == XB) {
                for (int ci = 0; ci < cin; ++ci) {
                  const float *wr = wp + (size_t)ci * 32;
                  for (int b = 0; b < XB; ++b) {
                    const float a = ip[b * cin + ci];
#pragma omp simd
                    for (int co = 0; co < 32; ++co) acc[b][co] += a * wr[co];
                  }
                }
              } else {
                for (int ci = 0; ci < cin; ++ci) {
                  const float *wr = wp + (size_t)ci * cout;
                  for (int b = 0; b < nb; ++b) {
                    const float a = ip[b * cin + ci];
                    for (int co = 0; co < cout; ++co) acc[b][co] += a * wr[co];
                  }
                }
              }
            }
        for (int b = 0; b < nb; ++b) {
          size_t o = (((size_t)z * Y + y) * X + x0 + b) * cout;
          for (int co = 0; co < cout; ++co) {
            float v = acc[b][co] + bias[co];
            if (relu_out && v < 0.0f) v = 0.0f;
            if (skip) v += skip[o + co];
            out[o + co] = v;
          }
        }
      }
    }
  }
  free(pad);
}

/* Returns 0 on success.
 * stop_after < 0: full forward, logits_out[n][Z][Y][X] = seed + update.
 * stop_after = k >= 0: dump the activation tensor [Z][Y][X][F] of batch item 0
 *   after conv number k into act_out (k = 0: conv0_a (post ReLU), 1: conv0_b,
 *   2i: conv{i}_a (post ReLU), 2i+1: conv{i}_b + skip) and return. */
int ffn_oracle_forward(const float *image, const float *seed, int n, int Z,
                       int Y, int X, int depth, int features,
                       const float *weights, float *logits_out, int stop_after,
                       float *act_out) {
  const int F = features;
  if (F > MAXF || depth < 1) return 1;
  const size_t V = (size_t)Z * Y * X;
  float *in2 = (float *)malloc(V * 2 * sizeof(float));
  float *a = (float *)malloc(V * F * sizeof(float));
  float *b = (float *)malloc(V * F * sizeof(float));
  if (!in2 || !a || !b) return 2;

  for (int item = 0; item < n; ++item) {
    const float *img = image + item * V;
    const float *sd = seed + item * V;
    /* tf.concat([patches, seed], 4): channel 0 = image, 1 = seed. */
    for (size_t v = 0; v < V; ++v) {
      in2[2 * v] = img[v];
      in2[2 * v + 1] = sd[v];
    }
    const float *w = weights;
    int k = 0;
    /* conv0_a: 2 -> F, ReLU */
    conv3x3x3(in2, a, NULL, w, w + 27 * 2 * F, Z, Y, X, 2, F, 0, 1);
    w += 27 * 2 * F + F;
    if (item == 0 && stop_after == k) { memcpy(act_out, a, V * F * sizeof(float)); goto done; }
    ++k;
    /* conv0_b: F -> F, no activation.  net = b */
    conv3x3x3(a, b, NULL, w, w + 27 * F * F, Z, Y, X, F, F, 0, 0);
    w += 27 * F * F + F;
    if (item == 0 && stop_after == k) { memcpy(act_out, b, V * F * sizeof(float)); goto done; }
    ++k;
    for (int i = 1; i < depth; ++i) {
      /* in_net = b ; a = relu(conv_a(relu(b))) ; b = conv_b(a) + in_net */
      conv3x3x3(b, a, NULL, w, w + 27 * F * F, Z, Y, X, F, F, 1, 1);
      w += 27 * F * F + F;
      if (item == 0 && stop_after == k) { memcpy(act_out, a, V * F * sizeof(float)); goto done; }
      ++k;
      conv3x3x3(a, b, b, w, w + 27 * F * F, Z, Y, X, F, F, 0, 0);
      w += 27 * F * F + F;
      if (item == 0 && stop_after == k) { memcpy(act_out, b, V * F * sizeof(float)); goto done; }
      ++k;
    }
    /* relu -> conv_lom (1x1x1, F -> 1) + bias ; logits = seed + update */
    float *lo = logits_out + item * V;
#pragma omp parallel for schedule(static)
    for (size_t v = 0; v < V; ++v) {
      float acc = 0.0f;
      const float *p = b + v * F;
      for (int c = 0; c < F; ++c) {
        float t = p[c] < 0.0f ? 0.0f : p[c];
        acc += t * w[c];
      }
      lo[v] = sd[v] + (acc + w[F]);
    }
  }
done:
  free(in2);
  free(a);
  free(b);
  return 0;
}

size_t ffn_oracle_weight_count(int depth, int features) {
  size_t F = (size_t)features;
  return 27 * 2 * F + F + (size_t)(2 * depth - 1) * (27 * F * F + F) + F + 1;
}

/* Number of OpenMP threads used by the convs (0 = leave unchanged).  Returns the
 * current maximum. */
int ffn_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
