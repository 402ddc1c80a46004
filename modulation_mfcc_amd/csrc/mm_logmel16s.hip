// libmodmfcc: the staged-sample fused kernel of the n_fft = 512 family -- frame + Hann + rFFT-512 + |.|^2 + mel + log
// (+ DCT, + clamp fix-up and trajectory rFFT in clip mode) for hop <= 252 -- the kernel BASELINE configs[1] / [2] / [4]
// and the reference's own default call run on.  Its own translation unit: 72 instantiations.  gfx950 only.
#include "mm_common.h"
#include "mm_s16.h"
#include "mm_fft16.hip.inc"          // f16:: register radix-16 core, Logmel512Params, the P-tile layout
#include "mm_logmel16w.hip.inc"      // lane-record layout (MM_W16_LT_PITCH), w16_read16
#include "mm_wpf_core.hip.inc"       // the 2048-point register transform of the clip-mode tail (n_mod 2048)
#include "mm_logmel16s.hip.inc"
