// Device-side signal segmentation.
#pragma once
#include "nrv_common.h"

namespace nrv {

// ---------------------------------------------------------------------------------------
// Device-side signal segmentation (preprocessing.py:103-131 through
// hoststage.segment_windows_f32): per base the 50 samples [st-25, st+25) clipped to the read,
// (x - shift)/scale in IEEE f64 then rounded to f32, symmetric zero padding (the odd sample goes in
// front).  Integer / exact work: the output is bit-identical to the host stage, which is pinned to
// the reference's own function.  One thread per output sample; stores are fully coalesced, the
// int16 gathers hit a ~100-byte neighbourhood per event.
// ---------------------------------------------------------------------------------------
struct SegRead {            // = nrv_read_desc (include/nanorev.h)
  long long raw_off, raw_len, ev_off, ev_len;
  double shift, scale;
};
struct SegArgs {
  const short* raw;         // all reads' samples, concatenated
  const int* starts;        // [N] event starts, relative to the start of their own read's samples
  const SegRead* reads;
  int n_reads;
  long long ev0;            // first event of this launch
  int n_ev;
  float* out;               // [n_ev][50]
};
__global__ void __launch_bounds__(256) segment_kernel(const SegArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.n_ev * 50) return;
  const int e = (int)(idx / 50), j = (int)(idx % 50);
  const long long E = a.ev0 + e;
  int lo_r = 0, hi_r = a.n_reads - 1;                   // last read with ev_off <= E
  while (lo_r < hi_r) {
    const int mid = (lo_r + hi_r + 1) >> 1;
    if (a.reads[mid].ev_off <= E) lo_r = mid; else hi_r = mid - 1;
  }
  const SegRead rd = a.reads[lo_r];
  float v = 0.f;
  if (E < rd.ev_off + rd.ev_len) {
    const long long st = a.starts[E], L = rd.raw_len;
    const long long lo = st - 25 <= 0 ? 0 : st - 25;
    const long long hi = st + 25 >= L ? L : st + 25;
    const long long seg = hi - lo, pad = 50 - seg;
    const long long left = pad > 0 ? pad / 2 + pad % 2 : 0;
    if (j >= left && j < left + seg)
      v = (float)(((double)a.raw[rd.raw_off + lo + j - left] - rd.shift) / rd.scale);
  }
  a.out[idx] = v;
}


}  // namespace nrv
