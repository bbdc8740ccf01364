/* nanorev_host.h - host-side helpers of the read path (libnanorev_host.so: plain C, no HIP, no GPU).
 *
 * The engine's callers cut a read into per-base events and compute each event's raw mean / standard deviation on
 * the host before the device call (the reference: nanorevutils/preprocessing.py:134-137, np.mean / np.std of
 * raw_signal[start:end] per base, inside signal_segmentation :85-170).  nrvh_event_stats is that loop in C with
 * NumPy's own summation order, so that the numbers are the ones NumPy gives, bit for bit
 * (tests/test_hoststage_golden.py compares it with np.mean / np.std event by event).
 */
#ifndef NANOREV_HOST_H
#define NANOREV_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces the per-base np.mean / np.std(ddof=0) of preprocessing.py:134-137.
 *   raw [n_raw]   the read's DAQ samples from its first event on (NanoReviser.py:120)
 *   starts [n_ev] event starts relative to raw[0], ascending; event i covers [starts[i], starts[i+1]), the last one
 *                 [starts[n_ev-1], starts[n_ev-1] + last_dur); ranges are clipped to n_raw like a Python slice
 *   mean, std [n_ev] f64 out; NaN for an empty range (what NumPy returns for an empty slice).
 * Returns 0, or -1 on bad arguments.  Pure function, thread-safe. */
int nrvh_event_stats(const int16_t* raw, int64_t n_raw, const int32_t* starts, int64_t n_ev, int32_t last_dur,
                     double* mean, double* std);

/* ABI version of this header (1). */
int nrvh_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
