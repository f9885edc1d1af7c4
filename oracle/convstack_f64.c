/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * The forward pass of convstack_oracle.c (reference
 * ffn/training/models/convstack_3d.py:26-56, 83-95; ffn/training/model.py:168-183)
 * in DOUBLE precision throughout: f32 image / seed / weights widened exactly, every
 * product and sum in f64, the logits rounded to f32 once at the end.  It is the
 * arithmetic every f32 implementation of the stack approximates ("the trajectory
 * of the arithmetic"), used by tools/make_golden.py
 * --forward f64c to mint whole-volume reference runs in hours instead of days
 * (torch's f64 conv3d has no vectorised CPU path: 1.1 s per FoV on 8 cores; this
 * file: see the tool's log).  Summation order: taps in (kz, ky, kx) order, input
 * channels ascending, one fused multiply-add per product -- in f64 the order
 * moves the result by ~1e-16 relative, eight orders below f32's rounding.
 *
 * Built on request only, for the host it runs on (-march=native): never part of
 * libffn_oracle.so, never loaded on the GPU box.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double v8d __attribute__((vector_size(64)));
#define F 32
#define XB 6

/* out[v][co] = bias[co] + sum in_pad[v + tap][ci] * w[tap][ci][co]  (+ skip), SAME
 * padding through a zero halo; relu_in on the way into the halo copy. */
static void conv_f64(const double *in, double *out, const double *skip, const double *w,
                     const double *bias, int Z, int Y, int X, int cin, int relu_in,
                     int relu_out, double *pad, int *pad_cin) {
  const int Yp = Y + 2, Xp = X + 2 + XB;
  /* the halo is zeroed once per layout (cin): the interior is rewritten by every conv */
  if (*pad_cin != cin) memset(pad, 0, (size_t)(Z + 2) * Yp * Xp * cin * sizeof(double));
  *pad_cin = cin;
#pragma omp parallel for collapse(2) schedule(static)
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y) {
      const double *src = in + ((size_t)z * Y + y) * X * cin;
      double *dst = pad + (((size_t)(z + 1) * Yp + (y + 1)) * Xp + 1) * cin;
      for (int i = 0; i < X * cin; ++i) {
        const double a = src[i];
        dst[i] = (relu_in && a < 0.0) ? 0.0 : a;
      }
    }
#pragma omp parallel for collapse(2) schedule(static)
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x0 = 0; x0 < X; x0 += XB) {
        const int nb = (X - x0) < XB ? (X - x0) : XB;
        v8d acc[XB][4];
        for (int b = 0; b < XB; ++b)
          for (int j = 0; j < 4; ++j) acc[b][j] = *(const v8d *)(bias + 8 * j);
        for (int kz = 0; kz < 3; ++kz)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const double *ip =
                  pad + (((size_t)(z + kz) * Yp + (y + ky)) * Xp + (x0 + kx)) * cin;
              const double *wp = w + (size_t)((kz * 3 + ky) * 3 + kx) * cin * F;
              for (int ci = 0; ci < cin; ++ci) {
                const v8d w0 = *(const v8d *)(wp + ci * F);
                const v8d w1 = *(const v8d *)(wp + ci * F + 8);
                const v8d w2 = *(const v8d *)(wp + ci * F + 16);
                const v8d w3 = *(const v8d *)(wp + ci * F + 24);
#pragma GCC unroll 6
                for (int b = 0; b < XB; ++b) {
                  const double s = ip[b * cin + ci];
                  const v8d a = {s, s, s, s, s, s, s, s};
                  acc[b][0] += a * w0;
                  acc[b][1] += a * w1;
                  acc[b][2] += a * w2;
                  acc[b][3] += a * w3;
                }
              }
            }
        for (int b = 0; b < nb; ++b) {
          const size_t o = (((size_t)z * Y + y) * X + x0 + b) * F;
          for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 8; ++c) {
              double v = acc[b][j][c];
              if (relu_out && v < 0.0) v = 0.0;
              if (skip) v += skip[o + 8 * j + c];
              out[o + 8 * j + c] = v;
            }
        }
      }
}

/* One FoV.  weights: the f32 blob of convstack_oracle.c (features = 32).  Returns 0. */
int ffn_oracle_forward_f64(const float *image, const float *seed, int Z, int Y, int X,
                           int depth, const float *weights, float *logits_out,
                           int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const size_t V = (size_t)Z * Y * X;
  const size_t padn = (size_t)(Z + 2) * (Y + 2) * (X + 2 + XB) * F;
  double *in2 = aligned_alloc(64, (V * 2 * sizeof(double) + 63) / 64 * 64);
  double *a = aligned_alloc(64, V * F * sizeof(double));
  double *b = aligned_alloc(64, V * F * sizeof(double));
  double *pad = aligned_alloc(64, padn * sizeof(double));
  const size_t wn = 27 * 2 * F + F + (size_t)(2 * depth - 1) * (27 * F * F + F) + F + 1;
  double *wd = aligned_alloc(64, (wn * sizeof(double) + 63) / 64 * 64 + 64 * 64);
  if (!in2 || !a || !b || !pad || !wd) return 2;
  /* every weight / bias block starts 64-byte aligned in wd */
  size_t off = 0, src = 0;
#define TAKE(N, DST)                                            \
  do {                                                          \
    DST = wd + off;                                             \
    for (size_t i = 0; i < (size_t)(N); ++i) wd[off + i] = (double)weights[src + i]; \
    src += (N);                                                 \
    off = (off + (N) + 7) / 8 * 8;                              \
  } while (0)
  for (size_t v = 0; v < V; ++v) {
    in2[2 * v] = (double)image[v];
    in2[2 * v + 1] = (double)seed[v];
  }
  double *w, *bs;
  int pad_cin = 0;
  TAKE(27 * 2 * F, w);
  TAKE(F, bs);
  conv_f64(in2, a, NULL, w, bs, Z, Y, X, 2, 0, 1, pad, &pad_cin);
  TAKE(27 * F * F, w);
  TAKE(F, bs);
  conv_f64(a, b, NULL, w, bs, Z, Y, X, F, 0, 0, pad, &pad_cin);
  for (int i = 1; i < depth; ++i) {
    TAKE(27 * F * F, w);
    TAKE(F, bs);
    conv_f64(b, a, NULL, w, bs, Z, Y, X, F, 1, 1, pad, &pad_cin);
    TAKE(27 * F * F, w);
    TAKE(F, bs);
    conv_f64(a, b, b, w, bs, Z, Y, X, F, 0, 0, pad, &pad_cin);
  }
  TAKE(F + 1, w);
#pragma omp parallel for schedule(static)
  for (size_t v = 0; v < V; ++v) {
    double acc = 0.0;
    const double *p = b + v * F;
    for (int c = 0; c < F; ++c) acc += (p[c] < 0.0 ? 0.0 : p[c]) * w[c];
    logits_out[v] = (float)((double)seed[v] + (acc + w[F]));
  }
  free(in2);
  free(a);
  free(b);
  free(pad);
  free(wd);
  return 0;
}
