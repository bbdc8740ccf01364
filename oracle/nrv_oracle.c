/* ORACLE - TEST INFRASTRUCTURE ONLY.  Plain-C f32 restatement of the reviser graph.
 *
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load this library, and only as the checker / reported CPU baseline.  The product
 * (libnanorev_hip.so) never links or calls it and has no CPU fallback.
 *
 * PARITY STATUS: parity unpinned for the model graph (see oracle/nrv_oracle.py header): the
 * reference's arithmetic lives in keras 2.2.4 / tensorflow 1.12, absent here; this file restates
 * the Keras-2.2.4 layer semantics at the reference's call sites and is itself checked against the
 * NumPy fp64 restatement and the committed goldens in tests/.
 *
 * Reference call sites followed:
 *   nanorevutils/nanorevcnn.py:17-26   Conv1d_BN  (Conv1D k=3 same relu, then BatchNorm)
 *   nanorevutils/nanorevcnn.py:29-38   identity_Block (two Conv1d_BN + Add with the input)
 *   nanorevutils/output_handeler.py:209-215  Dropout(identity), TD Flatten, TD Dense(64)
 *   nanorevutils/output_handeler.py:217-225  four Bidirectional(LSTM) + three BatchNorm, concat
 *   nanorevutils/output_handeler.py:230-237  Dense128/32/6 relu, Flatten, Dense16 relu, softmax
 *   (model2: :258-307, identical but 5 classes)
 *
 * Scalar, one window at a time; OpenMP over windows.  Sums are plain left-to-right f32
 * (compiled with -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXT 32
#define BN_EPS 1e-3f

typedef struct {
  const float *w[60];
  int T, C, act;
} model_t;

static size_t bind(model_t *m, const float *blob, int T, int C) {
  size_t sz[60];
  int i = 0;
#define BN(c) sz[i++] = c; sz[i++] = c; sz[i++] = c; sz[i++] = c
#define BIL(d, h) sz[i++] = (size_t)d * 4 * h; sz[i++] = (size_t)h * 4 * h; sz[i++] = 4 * h; \
                  sz[i++] = (size_t)d * 4 * h; sz[i++] = (size_t)h * 4 * h; sz[i++] = 4 * h
  sz[i++] = 24; sz[i++] = 8; BN(8);
  sz[i++] = 192; sz[i++] = 8; BN(8);
  BIL(6, 16); BN(32);
  BIL(32, 64); BN(128);
  sz[i++] = 400 * 64; sz[i++] = 64;
  BIL(192, 128); BN(256);
  BIL(256, 64);
  sz[i++] = 128 * 128; sz[i++] = 128; sz[i++] = 128 * 32; sz[i++] = 32; sz[i++] = 32 * 6; sz[i++] = 6;
  sz[i++] = (size_t)6 * T * 16; sz[i++] = 16; sz[i++] = (size_t)16 * C; sz[i++] = C;
  size_t off = 0;
  for (i = 0; i < 60; ++i) { m->w[i] = blob + off; off += sz[i]; }
  m->T = T; m->C = C;
  return off;
}

int64_t nrvo_n_params(int T, int C) {
  model_t m;
  static const float dummy[1] = {0};
  return (int64_t)bind(&m, dummy, T, C);
}

static inline float hsig(float z) { float v = z * 0.2f + 0.5f; return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
static inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }

static void bn(float *x, int n, const float *g, const float *b, const float *mu, const float *var) {
  for (int i = 0; i < n; ++i) {
    float inv = g[i] / sqrtf(var[i] + BN_EPS);
    x[i] = x[i] * inv + (b[i] - mu[i] * inv);
  }
}

/* signal branch for one event: sig[50] -> out[64] */
static void signal_branch(const model_t *m, const float *sig, float *out) {
  const float *const *w = m->w;
  float y1[50][8], y2[50][8];
  for (int p = 0; p < 50; ++p) {
    for (int o = 0; o < 8; ++o) {
      float v = w[1][o];
      for (int k = 0; k < 3; ++k) {
        int q = p + k - 1;
        if (q >= 0 && q < 50) v += sig[q] * w[0][k * 8 + o];
      }
      y1[p][o] = v < 0.f ? 0.f : v;          /* NaN in, NaN out (as np.maximum) */
    }
    bn(y1[p], 8, w[2], w[3], w[4], w[5]);
  }
  for (int p = 0; p < 50; ++p) {
    for (int o = 0; o < 8; ++o) {
      float v = w[7][o];
      for (int k = 0; k < 3; ++k) {
        int q = p + k - 1;
        if (q >= 0 && q < 50)
          for (int c = 0; c < 8; ++c) v += y1[q][c] * w[6][(k * 8 + c) * 8 + o];
      }
      y2[p][o] = v < 0.f ? 0.f : v;
    }
    bn(y2[p], 8, w[8], w[9], w[10], w[11]);
    for (int o = 0; o < 8; ++o) y2[p][o] += sig[p];          /* Add(): broadcast over channels */
  }
  const float *flat = &y2[0][0];                              /* index p*8+o */
  for (int j = 0; j < 64; ++j) out[j] = w[33][j];
  for (int k = 0; k < 400; ++k) {
    float a = flat[k];
    const float *wr = w[32] + (size_t)k * 64;
    for (int j = 0; j < 64; ++j) out[j] += a * wr[j];
  }
}

/* one direction of a Keras LSTM over T steps; x [T][D] -> out[t][ooff .. ooff+H) of stride ostr */
static void lstm_dir(const float *x, int T, int D, int H, const float *W, const float *U, const float *b,
                     int reverse, int act, float *out, int ostr, int ooff) {
  float h[128], c[128], z[512];
  memset(h, 0, sizeof h);
  memset(c, 0, sizeof c);
  for (int s = 0; s < T; ++s) {
    int t = reverse ? T - 1 - s : s;
    for (int j = 0; j < 4 * H; ++j) z[j] = 0.f;
    const float *xt = x + (size_t)t * D;
    for (int k = 0; k < D; ++k) {
      float a = xt[k];
      const float *wr = W + (size_t)k * 4 * H;
      for (int j = 0; j < 4 * H; ++j) z[j] += a * wr[j];
    }
    float zu[512];
    for (int j = 0; j < 4 * H; ++j) zu[j] = 0.f;
    for (int k = 0; k < H; ++k) {
      float a = h[k];
      const float *ur = U + (size_t)k * 4 * H;
      for (int j = 0; j < 4 * H; ++j) zu[j] += a * ur[j];
    }
    for (int j = 0; j < 4 * H; ++j) z[j] = (z[j] + zu[j]) + b[j];
    for (int j = 0; j < H; ++j) {
      float ig = act ? sigm(z[j]) : hsig(z[j]);
      float fg = act ? sigm(z[H + j]) : hsig(z[H + j]);
      float gg = tanhf(z[2 * H + j]);
      float og = act ? sigm(z[3 * H + j]) : hsig(z[3 * H + j]);
      c[j] = fg * c[j] + ig * gg;
      h[j] = og * tanhf(c[j]);
      out[(size_t)t * ostr + ooff + j] = h[j];
    }
  }
}

static void bilstm(const model_t *m, int base, const float *x, int T, int D, int H, float *out) {
  const float *const *w = m->w;
  lstm_dir(x, T, D, H, w[base], w[base + 1], w[base + 2], 0, m->act, out, 2 * H, 0);
  lstm_dir(x, T, D, H, w[base + 3], w[base + 4], w[base + 5], 1, m->act, out, 2 * H, H);
}

static void dense(const float *x, int K, int N, const float *W, const float *b, int relu, float *out) {
  for (int j = 0; j < N; ++j) out[j] = 0.f;
  for (int k = 0; k < K; ++k) {
    float a = x[k];
    const float *wr = W + (size_t)k * N;
    for (int j = 0; j < N; ++j) out[j] += a * wr[j];
  }
  for (int j = 0; j < N; ++j) {
    float v = out[j] + b[j];
    out[j] = relu ? (v < 0.f ? 0.f : v) : v;
  }
}

/* one window; sig [T][50] (or NULL with sig_out [T][64] given), read [T][6] -> prob[C] */
static void window(const model_t *m, const float *sig, const float *sig_out, const float *read, float *prob) {
  const float *const *w = m->w;
  const int T = m->T, C = m->C;
  float r1[MAXT * 32], r2[MAXT * 128], x3[MAXT * 192], r3[MAXT * 256], r4[MAXT * 128];
  bilstm(m, 12, read, T, 6, 16, r1);
  for (int t = 0; t < T; ++t) bn(r1 + t * 32, 32, w[18], w[19], w[20], w[21]);
  bilstm(m, 22, r1, T, 32, 64, r2);
  for (int t = 0; t < T; ++t) bn(r2 + t * 128, 128, w[28], w[29], w[30], w[31]);
  for (int t = 0; t < T; ++t) {
    memcpy(x3 + t * 192, r2 + t * 128, 128 * sizeof(float));       /* [read 128 | signal 64] */
    if (sig_out) memcpy(x3 + t * 192 + 128, sig_out + t * 64, 64 * sizeof(float));
    else signal_branch(m, sig + t * 50, x3 + t * 192 + 128);
  }
  bilstm(m, 34, x3, T, 192, 128, r3);
  for (int t = 0; t < T; ++t) bn(r3 + t * 256, 256, w[40], w[41], w[42], w[43]);
  bilstm(m, 44, r3, T, 256, 64, r4);
  float flat[MAXT * 6], d1[128], d2[32];
  for (int t = 0; t < T; ++t) {
    dense(r4 + t * 128, 128, 128, w[50], w[51], 1, d1);
    dense(d1, 128, 32, w[52], w[53], 1, d2);
    dense(d2, 32, 6, w[54], w[55], 1, flat + t * 6);
  }
  float feat[16], logit[8];
  dense(flat, 6 * T, 16, w[56], w[57], 1, feat);
  dense(feat, 16, C, w[58], w[59], 0, logit);
  float mx = logit[0];
  for (int c = 1; c < C; ++c) mx = logit[c] > mx ? logit[c] : mx;
  float e[8], sum = 0.f;
  for (int c = 0; c < C; ++c) { e[c] = expf(logit[c] - mx); sum += e[c]; }
  for (int c = 0; c < C; ++c) prob[c] = e[c] / sum;
}

/* n independent windows.  signal [n][T][50], read [n][T][6] -> prob [n][C], argmax [n].
 * Returns 0, or -1 on bad arguments. */
int nrvo_predict(const float *blob, int64_t n_f32, int T, int C, int act, const float *signal,
                 const float *read, int64_t n, float *prob, int8_t *argmax, int threads) {
  if (T < 1 || T > MAXT || (C != 5 && C != 6) || n_f32 != nrvo_n_params(T, C)) return -1;
  model_t m;
  bind(&m, blob, T, C);
  m.act = act;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < n; ++i) {
    float p[8];
    window(&m, signal + (size_t)i * T * 50, (const float *)0, read + (size_t)i * T * 6, p);
    int best = 0;
    for (int c = 0; c < C; ++c) {
      if (prob) prob[(size_t)i * C + c] = p[c];
      if (p[c] > p[best]) best = c;
    }
    if (argmax) argmax[i] = (int8_t)best;
  }
  return 0;
}

/* whole read: per-event arrays, sliding windows i in [0, N-T), signal branch once per event */
int nrvo_predict_read(const float *blob, int64_t n_f32, int T, int C, int act, const float *sig_ev,
                      const float *feat_ev, int64_t N, float *prob, int8_t *argmax, int threads) {
  if (T < 1 || T > MAXT || (C != 5 && C != 6) || n_f32 != nrvo_n_params(T, C)) return -1;
  model_t m;
  bind(&m, blob, T, C);
  m.act = act;
  int64_t n = N - T;
  if (n <= 0) return 0;
  float *so = (float *)malloc((size_t)N * 64 * sizeof(float));
  if (!so) return -2;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 32) num_threads(threads > 0 ? threads : 1)
  for (int64_t e = 0; e < N; ++e) signal_branch(&m, sig_ev + (size_t)e * 50, so + (size_t)e * 64);
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < n; ++i) {
    float p[8];
    window(&m, (const float *)0, so + (size_t)i * 64, feat_ev + (size_t)i * 6, p);
    int best = 0;
    for (int c = 0; c < C; ++c) {
      if (prob) prob[(size_t)i * C + c] = p[c];
      if (p[c] > p[best]) best = c;
    }
    if (argmax) argmax[i] = (int8_t)best;
  }
  free(so);
  return 0;
}
