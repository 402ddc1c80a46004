// The staged-sample n_fft = 512 kernel (mm_logmel16s.hip.inc) is its own translation unit (72 instantiations): this
// header carries what the plan / dispatch code in mm_api.hip needs from it -- the LDS layout and the launch wrappers.
#pragma once
#include "mm_common.h"

#define MM_S16_S_OFF (MM_LM_P_FLOATS * 4)
#ifndef MM_S16F_CH
#define MM_S16F_CH 12     // DCT steps (4 filters each) whose operands are fetched together
#endif
#ifndef MM_S16F_W
#define MM_S16F_W 0.5     // mel share of a DCT wave relative to the other waves
#endif
#ifndef MM_S16_TW_REG
#define MM_S16_TW_REG 0    // stage-1 twiddles kept in registers (even, <= 14)
#endif
#ifndef MM_S16_WP_REG
#define MM_S16_WP_REG 1    // split twiddles kept in registers
#endif
#define MM_S16_LT_OFF(NR) (MM_S16_S_OFF + (NR) * 16384)
#define MM_S16_TAB_OFF(NR) (MM_S16_LT_OFF(NR) + 16 * MM_W16_LT_PITCH * 4)

#define MM_S16_CPW_MAX 32         // clip mode: at most this many clips per workgroup (extreme slots in LDS, 128 B each)
#define MM_S16_FIN_REC 24
#define MM_S16_FIN_TAB_BYTES ((32 * MM_S16_FIN_REC + 15) * 8)
#define MM_S16_FIN_TAB_OFF (16 * 9216)
// clip mode, plans with empty filters: the per-clip add factors (a float per clip of the workgroup) lie behind the extreme
// slots of the launch's largest workgroup (clips_max x 128 B) and, at n_mod 2048, behind the 2048-point tail's buffers
#define MM_S16_DELTA_OFF(red_off, clips_max, n_mod)                                                             \
  (((n_mod) == 2048 && (red_off) + (clips_max) * 128 < MM_S16_FIN2K_BYTES) ? MM_S16_FIN2K_BYTES : (red_off) + (clips_max) * 128)
#define MM_S16_FIN2K_BYTES 103424   // n_mod 2048: lane table of the 2048-point transform (64 x 116 floats) + 16 x 4.5 KB exchange buffers

struct Logmel512Params;
// mode 0 power rows | 1 log-mel (+ fused DCT) | (1 with q.out_mod set ->) 2 clip mode; nr = 16-byte staging groups per thread
void launch_s16(int mode, int nr, bool pre, bool odd, bool unal, dim3 grid, size_t lds, hipStream_t st,
                const Logmel512Params& q);
bool set_s16_attr(int bytes);       // hipFuncAttributeMaxDynamicSharedMemorySize on every instantiation (current device)
