// nrv_kernels.h - gfx950 (MI355X / CDNA4) device kernels of the NanoReviser window reviser.
//
// What is computed: the two Keras graphs of the reference,
//   nanorevutils/output_handeler.py:206-255 (model1) and :258-307 (model2),
//   CNN block nanorevutils/nanorevcnn.py:17-38,
// with Keras-2.2.4 inference semantics.  Results are IEEE-f32 grade.
//
// How it is mapped (MI355X-first, not a translation of TF ops):
//   * Contractions run on the matrix pipe with f32-grade products: by default operands scaled by a
//     static power of two and split into two f16 terms (three products per f32 product, f32
//     accumulation; nrv_lstm_f16x2.h, nrv_cnn_r.h, nrv_head_f16x2.h); or as an exact three-term
//     bf16 split on v_mfma_f32_32x32x16_bf16 (six products; nrv_lstm_bf16x3.h, head_mlp_split_kernel);
//     or on v_mfma_f32_32x32x2_f32 / 16x16x4 (NRV_PREC_F32, and always for the small layers).
//   * Activations between kernels live in an MFMA-native tiled layout
//         act[tile32][t][kq][32 rows][4]      (kq = feature/4)
//     so that one wave-wide 16-byte load IS an A fragment and is a single contiguous 1 KiB request.
//     Weights are pre-packed on the host into the matching B-fragment order (f32 and split-bf16
//     forms), so B operands stream L2 -> VGPR with no LDS staging.
//   * FIVE launches per group of 4096 windows in the default (f16x2) mode:
//       cnn_r_kernel      signal branch (conv1 on the VALU -> conv2 -> dense 400->64 on the matrix pipe, in registers)
//                         + the 6->16 Bi-LSTM (lstm1_unit) as four more waves of the same workgroups
//       lstm2_u_kernel    32->64 Bi-LSTM, transposed products, two waves per 16-row chain (r05)
//       lstm_h2w_kernel   192->128 Bi-LSTM, 16x16x32 f16 tiles, eight waves: two per SIMD running a step in opposite order
//       lstm_h2s_kernel   256->64 Bi-LSTM <64,0,64,...>, 16x16x32 f16 tiles, one wave per SIMD
//       head_h2_kernel    per-timestep MLP 128->128->32->6 + flatten, feature dense, softmax, argmax
//     seven in the bf16x3 / f32 modes (cnn_kernel, lstm1_kernel, lstm_pair_kernel / lstm_split_kernel or
//     lstm_layer_kernel x3, head_mlp(_split)_kernel, head_final_kernel); plus segment_kernel when reads arrive as raw
//     samples.  Rows (windows) are independent: no inter-workgroup communication anywhere.
//   * One Bi-LSTM layer = one launch; a wave owns a group of hidden units x 4 gates x R row tiles, so
//     i,f,g,o of one (window, unit) sit in the same lane and the cell update is register-local; c never leaves the
//     wave, h_t goes through an LDS image (lstm_h2s_kernel: double-buffered, one barrier per step; lstm_h2w_kernel:
//     one image, two barriers; lstm2_u_kernel: the two halves of a chain's h_t through LDS, one barrier)
//     and is written out coalesced, the BatchNorm behind it fused or folded into the next layer's weights.
//
// Compile-time switches left in these sources (round 6 removed the experiment variants that lost; HISTORY.md keeps the record).
// None of them selects another product kernel; all are 0 / undefined in the product build (__graft_entry__.build_hip):
//   NRV_STAMP (+ NRV_STAMP_REC_ENTRIES)   DIAGNOSTIC ONLY: s_memtime stamps at the phase edges of lstm_h2s_kernel / lstm_h2w_kernel and
//                                         nrv_exp_only_stage / nrv_exp_stamps (scripts/gpu_stamps*.py, gpu_power_stage.py)
//   NRV_DEV_FAST (-> NRV_ACT1)            DEVELOPMENT ONLY: f16x2 mode with hard_sigmoid only, half the compile time (tools/lstm_exp.sh)
//   NRV_L3_WS_NBG                         the 192->128 layer's weight ring (4; 8 spills: r04w) - bench.KERNEL_SIGNATURE follows it
// Run-time environment variables of the engine: NRV_COALESCE / NRV_LANES (round 3's stream lanes instead of coalesced 4096-window
// groups: tests/test_gpu_parity.py grouping tests), NRV_READ_STAGE, NRV_WINDOW_STAGE_MAX, NRV_HOST_REGISTER, NRV_HOST_TRACE.
#pragma once
#include "nrv_common.h"        // vector types, buffer loads, activations, ActView
#include "nrv_cnn.h"           // cnn_kernel
#include "nrv_lstm1.h"         // lstm1_kernel
#include "nrv_lstm_f32.h"      // lstm_layer_kernel, lstm_block / lstm_grid
#include "nrv_lstm_bf16x3.h"   // lstm_split_kernel, lstm_pair_kernel
#include "nrv_lstm_f16x2.h"    // the scaled two-term f16 split (NRV_PREC_F16X2): types, split2, LstmH2Args
#include "nrv_lstm_f16x2s.h"   // lstm_h2s_kernel (256->64 layer of the f16x2 mode, 16x16x32 tiles, one wave per SIMD)
#include "nrv_lstm_f16x2w.h"   // lstm_h2w_kernel (192->128 layer: eight waves, two groups running a step in opposite order)
#include "nrv_lstm2_u.h"       // lstm2_u_kernel (32->64 layer: transposed products, two waves per chain, h halves through LDS: r05)
#include "nrv_cnn_r.h"         // cnn_r_kernel (signal branch of the f16x2 mode: conv1 -> conv2 -> dense in registers)
#include "nrv_head.h"          // head_mlp_kernel, head_mlp_split_kernel, head_final_kernel
#include "nrv_head_f16x2.h"    // head_h2_kernel (per-timestep MLP + per-window tail, f16x2 mode)
#include "nrv_segment.h"       // segment_kernel
