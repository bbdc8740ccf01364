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

/* One single-read fast5 file -> the arrays of one read as the device call takes them (nrv_predict_reads_raw,
 * include/nanorev.h): what the reference's get_read_data (nanorevutils/nanorev_fast5_handeler.py:39-150), the medians
 * and per-base statistics of signal_segmentation (nanorevutils/preprocessing.py:100-101, 134-137) and the feature
 * rows of nanorevtrainutils.py:162-169 produce, read with an HDF5 subset written from the format specification
 * (csrc/nrv_host_fast5.c).  The Python host stage (h5lite.py + hoststage.py) is the definition: this gives the same
 * numbers bit for bit or declines.
 * All arrays are malloc'ed by the call and released by nrvh_free_read. */
typedef struct {
  int64_t n_raw;       /* samples from the read's first event on */
  int64_t n_ev;        /* bases */
  int16_t* raw;        /* [n_raw] */
  int32_t* starts;     /* [n_ev] event starts relative to raw[0] */
  float* feat;         /* [n_ev][6] color/300, mean/shift, std/scale, length/10, Albacore mean, Albacore stdv */
  char* bases;         /* [n_ev] + NUL: the basecalls (model_state letters) */
  double shift, scale; /* median and median absolute deviation of raw[] */
  char* fastq;         /* the file's Fastq record (NUL-terminated; NULL when absent or not asked for) */
  int64_t fastq_len;
} nrvh_read;

#define NRVH_OK 0
#define NRVH_UNSUPPORTED 1   /* not the subset this reader knows (pre-versioned basecaller output, another HDF5 layout,
                                a missing group, ...): run the Python host stage, which decides and words the error */
#define NRVH_E_READ 2        /* the reference's own failure for this file ("Events is too short ...", "Signal is shorter
                                than the Events"): the Python path raises the same */
#define NRVH_E_IO 3
#define NRVH_E_ARG 4

/* Returns NRVH_OK or one of the codes above with a short reason in err[err_len].  want_fastq != 0: also read the Fastq
 * record.  Thread-safe; touches no Python object (ctypes callers run it with the GIL released). */
int nrvh_load_fast5(const char* path, const char* group, const char* subgroup, int want_fastq, nrvh_read* out,
                    char* err, int err_len);
void nrvh_free_read(nrvh_read* r);

/* Several files -> the concatenated arrays of ONE device call (what cli._load_bundle builds): the reads that came
 * back NRVH_OK one behind the other in raw / starts / feat / bases, their (raw_len, ev_len, shift, scale) in meta, and
 * per file the status code with the reason for anything else (those files go through the Python host stage). */
#define NRVH_ERR_LEN 96
typedef struct {
  int32_t n_files, n_ok;
  int64_t n_raw, n_ev;   /* totals over the NRVH_OK reads */
  int16_t* raw;          /* [n_raw] */
  int32_t* starts;       /* [n_ev] relative to each read's own first sample */
  float* feat;           /* [n_ev][6] */
  char* bases;           /* [n_ev] + NUL */
  double* meta;          /* [n_files][4]: raw_len, ev_len, shift, scale (zeros where status != NRVH_OK) */
  int32_t* status;       /* [n_files] */
  char* fastq;           /* the Fastq records of the NRVH_OK reads one behind the other */
  int64_t* fastq_off;    /* [n_files + 1] offsets into fastq; -1: that file has no record */
  char* errors;          /* [n_files][NRVH_ERR_LEN] */
} nrvh_bundle;
int nrvh_load_bundle(const char* const* paths, int n, const char* group, const char* subgroup, int want_fastq,
                     nrvh_bundle* out);
void nrvh_free_bundle(nrvh_bundle* b);

/* The calls of one read -> the revised read -> its output file: the merge of nanorevutils/output_handeler.py:83, 104-122
 * (decode as SURVEY.md 8a a16: model1 class = label, model2 class k = label k + 1; window i revises base
 * i + (T - 1) / 2), the record of output_handeler.py:26-62 byte for byte, written to a temporary and renamed to dst.
 *   bases [n_ev] the read's basecalls; a1, a2 [n_win] argmax of model1 / model2; qc [n_win] one Phred character per
 *   window for FASTQ (fastq != 0) or NULL; name: the record's name (file name, blanks replaced by "|||").
 * n_written: characters of the revised sequence. */
int nrvh_finish_read(const char* bases, int64_t n_ev, const int8_t* a1, const int8_t* a2, int64_t n_win, int T,
                     const uint8_t* qc, const char* name, const char* dst, int fastq, int64_t* n_written);

/* nrvh_finish_read for every read of ONE device call: bases / a1 / a2 / qc are the call's concatenated arrays (read r
 * has ev_len[r] bases and max(ev_len[r] - T, 0) windows; its window i is window e0 + i of the call, e0 = the bases in
 * front of it; n_win_total windows in all).  status[r] / n_written[r] per read; a failing read does not stop the others. */
int nrvh_finish_bundle(const char* bases, const int64_t* ev_len, int n_reads, const int8_t* a1, const int8_t* a2,
                       int64_t n_win_total, int T, const uint8_t* qc, const char* const* names, const char* const* dsts,
                       int fastq, int64_t* n_written, int32_t* status);

/* ABI version of this header (2). */
int nrvh_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
