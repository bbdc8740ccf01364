/* Host-side helpers of the read path (include/nanorev_host.h).  Plain C, built with
 *   gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math
 * - the point of this file is to give NumPy's numbers bit for bit, so nothing here may be re-associated or fused. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nanorev_host.h"

int nrvh_abi_version(void) { return 2; }

/* NumPy's float64 add.reduce over a contiguous run (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum):
 * fewer than 8 values are added in order; up to 128 go through eight running sums combined as
 * ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) with the remainder added in order; longer runs are halved
 * (the first half rounded down to a multiple of 8) and the halves' sums added. */
static double pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

int nrvh_event_stats(const int16_t* raw, int64_t n_raw, const int32_t* starts, int64_t n_ev, int32_t last_dur,
                     double* mean, double* std) {
  if (n_raw < 0 || n_ev < 0 || (n_ev > 0 && (!starts || !mean || !std)) || (n_raw > 0 && !raw)) return -1;
  double stack_buf[256];
  double* buf = stack_buf;
  int64_t cap = 256;
  double* heap = 0;
  for (int64_t e = 0; e < n_ev; ++e) {
    int64_t s = starts[e], t = e + 1 < n_ev ? (int64_t)starts[e + 1] : (int64_t)starts[e] + last_dur;
    if (s < 0) s = 0;                       /* (a negative start does not occur; a slice would wrap, this clips) */
    if (s > n_raw) s = n_raw;
    if (t > n_raw) t = n_raw;
    const int64_t n = t - s;
    if (n <= 0) { mean[e] = NAN; std[e] = NAN; continue; }
    if (n > cap) {
      if (heap) free(heap);
      heap = (double*)malloc((size_t)(2 * n) * sizeof(double));
      if (!heap) return -1;
      buf = heap;
      cap = 2 * n;
    }
    /* np.mean: add.reduce / n.  np.std -> _var: arrmean = add.reduce / n; x = arr - arrmean; x *= x;
     * add.reduce(x) / n; sqrt (numpy/_core/_methods.py) */
    for (int64_t i = 0; i < n; ++i) buf[i] = (double)raw[s + i];
    const double m = pairwise_sum(buf, n) / (double)n;
    for (int64_t i = 0; i < n; ++i) {
      const double d = buf[i] - m;
      buf[i] = d * d;
    }
    mean[e] = m;
    std[e] = sqrt(pairwise_sum(buf, n) / (double)n);
  }
  if (heap) free(heap);
  return 0;
}
