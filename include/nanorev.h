/* nanorev.h - C-ABI of libnanorev_hip.so, the MI355X (gfx950) engine behind NanoReviser's
 * window reviser (model1 + model2).
 *
 * The reference has no FFI for this path; the seam it exposes is the pair of Keras callables
 * built by nanorevutils/output_handeler.py:206-255 (get_model1) and :258-307 (get_model2),
 *     Model(inputs=[signal_input (B,T,50,1) f32, read_input (B,T,6) f32], outputs=[(B,C) softmax])
 * (:250-251, :302-303), which NanoReviser.py:129-130 constructs per read and would call as
 * `model.predict([signal_x, read_x])`.  Each entry point below cites the reference interface
 * it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer adds on the reference
 * side.
 *
 * Conventions: plain pointers and sizes only.  Every function returns 0 on success and a
 * negative nrv_status otherwise; nothing throws.  The engine copies weights at create time and
 * never keeps a caller pointer past the return of a call.  Calls on ONE handle are not
 * re-entrant (one caller at a time: two threads inside the same handle corrupt whole launch
 * groups); SEVERAL handles per (process, GPU) may be driven by several threads at once
 * (scripts/gpu_two_engines.py; the device is shared, so this buys overlap, not throughput).  There is NO CPU fallback: without a usable HIP device
 * nrv_create fails with NRV_E_NO_DEVICE.
 */
#ifndef NANOREV_H
#define NANOREV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nrv_handle nrv_handle;

typedef enum {
  NRV_OK = 0,
  NRV_E_INVALID = -1,    /* bad argument (null pointer, negative size, unsupported T) */
  NRV_E_WEIGHTS = -2,    /* weight blob does not have the size the graph requires */
  NRV_E_NO_DEVICE = -3,  /* no HIP device / device index out of range */
  NRV_E_HIP = -4,        /* a HIP runtime call failed; see nrv_last_error */
  NRV_E_NOMEM = -5
} nrv_status;

#define NRV_BACKEND_HIP 1

/* Flat little-endian f32 blob holding the 60 tensors of one model in Keras' positional
 * `load_weights` order (what `model.load_weights(<S>_win13_50ep_model{1,2}.h5)` would read;
 * path convention NanoReviser.py:191-193; order SURVEY.md Appendix A-11). */
typedef struct {
  const float* data;
  int64_t n_f32;
} nrv_weights;

/* Replaces get_model1()/get_model2() + load_weights (output_handeler.py:206-307;
 * NanoReviser.py:129-130).  T = window length (SENT_LEN, output_handeler.py:201; the shipped
 * weights are T=11).  recurrent_act: 0 = hard_sigmoid (Keras 2.2.4 default, what the weights
 * were trained with), 1 = sigmoid (Keras >= 2.3 behaviour, enviroment/NanoReviser_macOS.yaml). */
int nrv_create(const nrv_weights* model1, const nrv_weights* model2, int T, int device,
               int recurrent_act, nrv_handle** out);

void nrv_destroy(nrv_handle* h);

/* Replaces model1.predict([signal, read]) and model2.predict([signal, read])
 * (output_handeler.py:250-251, :302-303) on n independent windows.
 *   signal [n][T][50] f32, read [n][T][6] f32 (feature order nanorevtrainutils.py:169).
 *   p1 [n][6], p2 [n][5] softmax outputs; a1, a2 [n] argmax (ties -> lowest index).
 * Any output pointer may be NULL.  All pointers are HOST memory. */
int nrv_predict(nrv_handle* h, const float* signal, const float* read, int64_t n,
                float* p1, float* p2, int8_t* a1, int8_t* a2);

/* Whole-read form: the sliding windows x[i:i+T], i in [0, N-T) of
 * nanorevtrainutils.py:198-209 are formed on the device and the signal branch runs once per
 * event instead of once per (window, timestep).
 *   sig_ev [N][50] f32, feat_ev [N][6] f32; outputs have N-T rows (0 rows if N <= T). */
int nrv_predict_read(nrv_handle* h, const float* sig_ev, const float* feat_ev, int64_t N,
                     float* p1, float* p2, int8_t* a1, int8_t* a2);

/* Whole reads from RAW samples: the signal segmentation of preprocessing.py:103-131 (per base the
 * 50 samples around its first sample, (x - shift)/scale, zero padded) runs on the device, so a read
 * crosses PCIe as int16 samples + int32 event starts + the 6 event features (~46 B per base instead
 * of 224).  Several reads can share one call: `raw` and the per-event arrays are the reads'
 * arrays concatenated, `reads[r]` says where read r lies in them and carries its shift / scale
 * (medians over the read, NanoReviser.py:120 -> preprocessing.py:99-100, computed by the caller).
 *   raw [n_raw] int16; starts [N] int32, relative to the first sample of their own read;
 *   feat_ev [N][6] f32.  Outputs: N - T rows, exactly those of nrv_predict_read on the
 *   concatenated per-event arrays (windows that straddle two reads are the caller's to skip). */
typedef struct {
  int64_t raw_off, raw_len;   /* samples of the read inside `raw` */
  int64_t ev_off, ev_len;     /* events of the read inside the per-event arrays */
  double shift, scale;
} nrv_read_desc;
int nrv_predict_reads_raw(nrv_handle* h, const int16_t* raw, int64_t n_raw, const int32_t* starts,
                          const float* feat_ev, int64_t N, const nrv_read_desc* reads, int n_reads,
                          float* p1, float* p2, int8_t* a1, int8_t* a2);
/* The same call in two halves (r06), so that a caller can have the device work on call k+1 while it collects call k - what the
 * reference's per-read Pool tasks (NanoReviser.py:203-219) get from running several `predict` calls side by side:
 *   nrv_reads_raw_begin  copies the inputs (92 B per base) into page-locked staging - they may be freed or reused as soon as it
 *                        returns -, uploads them in one transfer and enqueues EVERY stage of the call; *ticket names the call;
 *   nrv_reads_raw_end    waits for that call's results and writes them to the p1 / p2 / a1 / a2 given to _begin (which must stay
 *                        valid until then; any may be NULL).  A call that tripped the f16x2 range guard is re-run whole on the
 *                        f32 kernels here (nrv_saturated's *reruns counts it).
 * At most two calls in flight per handle (a third _begin returns NRV_E_INVALID); calls complete in the order they began; between
 * a _begin and its _end only these two entry points may be called on the handle.  nrv_predict_reads_raw IS _begin + _end. */
int nrv_reads_raw_begin(nrv_handle* h, const int16_t* raw, int64_t n_raw, const int32_t* starts,
                        const float* feat_ev, int64_t N, const nrv_read_desc* reads, int n_reads,
                        float* p1, float* p2, int8_t* a1, int8_t* a2, int* ticket);
int nrv_reads_raw_end(nrv_handle* h, int ticket);
/* The segmentation alone: sig_ev [N][50] f32 to HOST memory (what nrv_predict_reads_raw feeds the
 * signal branch; bit-identical to the host stage - used by the parity tests). */
int nrv_segment_reads(nrv_handle* h, const int16_t* raw, int64_t n_raw, const int32_t* starts, int64_t N,
                      const nrv_read_desc* reads, int n_reads, float* sig_ev);

/* Same two calls with DEVICE pointers, enqueued on the handle's stream without a host sync
 * (call nrv_sync, or synchronise the stream you passed to nrv_set_stream).  The inputs must be
 * complete in stream order.  The handle's own stream is a blocking stream, i.e. it is ordered
 * after work already queued on the legacy default stream (where e.g. torch produces tensors by
 * default); producers on any other stream must be synchronised by the caller, or the handle
 * pointed at that stream with nrv_set_stream. */
int nrv_predict_device(nrv_handle* h, const float* d_signal, const float* d_read, int64_t n,
                       float* d_p1, float* d_p2, int8_t* d_a1, int8_t* d_a2);
int nrv_predict_read_device(nrv_handle* h, const float* d_sig_ev, const float* d_feat_ev,
                            int64_t N, float* d_p1, float* d_p2, int8_t* d_a1, int8_t* d_a2);

/* Windows per launch group (Keras' predict(batch_size=...)); default 4096.  Results do not depend on the
 * grouping (every window is its own row of every kernel).  A group below 4096 windows would leave most of the
 * chip idle, so consecutive smaller groups of ONE call are coalesced into 4096-window launches (nrv_get_batch
 * still reports what was set); values above 4096 make larger launches (measured: no gain, the workspace
 * outgrows the Infinity Cache).  NRV_COALESCE=0 in the environment restores the uncoalesced form, in which
 * groups of <= 2048 windows (batch % 32 == 0) run concurrently on several streams. */
int nrv_set_batch(nrv_handle* h, int batch_windows);
int nrv_get_batch(nrv_handle* h);

/* Matrix arithmetic of the three large Bi-LSTM layers (32->64, 192->128, 256->64), the per-timestep
 * dense layers (128->128->32->6) and the signal branch's 400->64 layer: 98 % of the FLOPs.  All modes
 * accumulate in f32 and meet the same parity bars (every test of tests/test_gpu_parity.py runs once per
 * mode; DESIGN.md 5 has the figures over all 81 770 fixture windows: 0 argmax differences in any mode):
 *   NRV_PREC_F16X2 (default)  every operand is scaled by a power of two fixed at nrv_create from static
 *                    bounds, split into two f16 terms, and each product formed from three term pairs
 *                    on the f16 matrix pipe; activations travel between the kernels already split.
 *                    ~2.6x the f32 mode's throughput.
 *   NRV_PREC_BF16X3  every f32 operand is split exactly into three bf16 terms and each product formed
 *                    from the six term pairs that matter, on the bf16 matrix pipe; ~1.5x f32 mode;
 *   NRV_PREC_F32     plain f32 matrix instructions.
 * The convolutions, the first Bi-LSTM (6->16) and the per-window tail are f32 in every mode.
 * Range of NRV_PREC_F16X2: every activation has a static bound except the signal branch (the reference
 * normalises samples as (raw - median) / MAD without clipping, preprocessing.py:120-131, and computes in
 * f32, nanorevcnn.py:24-37, so spike samples and tiny MADs have a well-defined answer).  The f16x2 signal
 * branch represents |S| < 1023; a launch group that leaves that range (or carries a NaN / Inf sample)
 * is detected on the device and NEVER returned as is: see nrv_saturated.
 * Takes effect from the next call.  The environment variable NRV_PRECISION=f32|bf16x3|f16x2 sets the
 * mode a new handle starts in. */
#define NRV_PREC_F32 0
#define NRV_PREC_BF16X3 1
#define NRV_PREC_F16X2 2
int nrv_set_precision(nrv_handle* h, int mode);
int nrv_get_precision(nrv_handle* h);

/* Range guard of NRV_PREC_F16X2 (no-op in the other modes, which keep f32 buffers).
 *   Host-pointer entry points (nrv_predict, nrv_predict_read, nrv_predict_reads_raw): a pipeline stage whose
 *   signal branch left the f16 range is re-run on the NRV_PREC_F32 kernels before its results are handed over;
 *   *reruns counts such stages since nrv_create (informational - the results are already the f32 mode's).
 *   Device-pointer entry points (asynchronous): *pending is non-zero when a launch group enqueued since the
 *   previous nrv_saturated call left the range; the outputs of those calls must be discarded and the calls
 *   repeated after nrv_set_precision(h, NRV_PREC_F32) (engine.Reviser.predict_device_checked does that).
 *   Synchronises the handle's stream and clears the pending count.  Either pointer may be NULL. */
int nrv_saturated(nrv_handle* h, int64_t* pending, int64_t* reruns);

/* Use an existing hipStream_t (e.g. torch's current stream); NULL restores the handle's own. */
int nrv_set_stream(nrv_handle* h, void* hip_stream);
int nrv_sync(nrv_handle* h);

/* Per-kernel timing with HIP events recorded on the launch stream.  While enabled every
 * launch group is bracketed by events; nrv_prof_read synchronises, adds up the elapsed times
 * since the last read and returns, per kernel slot, total milliseconds and launch count.
 * Slots: 0 cnn, 1 lstm1, 2 lstm2, 3 lstm3, 4 lstm4, 5 head.  on = 1: every kernel (seven event
 * records per group, ~3 % of a 4096-window group); on = 2: only slot 3, the dominant kernel (two
 * records per group, ~12 us of idle pipe); on = 3: slot 3 on every 8th group only. */
#define NRV_N_KERNELS 6
int nrv_prof_enable(nrv_handle* h, int on);
int nrv_prof_read(nrv_handle* h, double* ms_total /*[NRV_N_KERNELS]*/,
                  int64_t* launches /*[NRV_N_KERNELS]*/);
const char* nrv_kernel_name(int slot);
/* Mean elapsed time, in microseconds, of an EMPTY bracket (two event records back to back on the handle's stream):
 * what a bracketed launch's figure contains beyond the kernel's own duration.  bench.py reports it next to the
 * bracketed figures so that they can be set against a rocprofv3 kernel trace of the same command. */
int nrv_prof_overhead(nrv_handle* h, double* us);

/* Message of the last failure on this handle (or, with h == NULL, of the last failed
 * nrv_create on this thread).  Never NULL. */
const char* nrv_last_error(nrv_handle* h);

/* Number of HIP devices visible to this process (0 when there is none or the runtime fails): what
 * the command line shards reads over (NanoReviser.py:214-219 sharded files over Pool workers). */
int nrv_device_count(void);

/* Always NRV_BACKEND_HIP: the library has no other backend. */
int nrv_backend(nrv_handle* h);

/* Window length the handle was created with. */
int nrv_window(nrv_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* NANOREV_H */
