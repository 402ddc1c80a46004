// Internal declarations shared by mm_tables.cpp (host tables) and the .hip translation units (device code + C ABI).
#ifndef MM_INTERNAL_H
#define MM_INTERNAL_H

#include <vector>

#include "../../include/modmfcc.h"

namespace mm {

struct MelCsr {
  std::vector<int> start, len, off;  // per filter: first bin, run length, offset into w
  std::vector<float> w;
};

// Sweep form of the mel matrix for the lane<->frame kernels: every bin k feeds at most the
// "falling" filter d[k] (weight wlo) and the "rising" filter d[k]+1 (weight whi), d monotone.
struct MelSweep {
  std::vector<float> wlo, whi;
  std::vector<int> d;
  std::vector<int> part;  // [n_waves][4] = {k_begin, k_end, m_begin, m_end}
};
// weights (optional, [n_waves]): relative share of the sweep cost each wave should get (default: equal)
bool build_mel_sweep(const mm_config& c, const float* dense, int n_waves, MelSweep* out, const double* weights = nullptr);

// Run form of the sweep for the fused kernel's phase B.  A "run" is the set of bins with one value
// of d; per wave the runs d = m_begin-1 .. m_end-1 are listed in order.  Bins are handled in
// aligned groups of 4 (one ds_read_b128 of the power row); weights outside the run are zero.
struct MelRuns {
  std::vector<int> hdr;     // [n_runs][4] = {first bin (multiple of 4), n_groups, first group, d}
  std::vector<float> grp;   // [n_groups][8] = {wlo x4, whi x4}
  std::vector<int> part;    // [n_waves][4] = {run_begin, run_end, m_begin, m_end}
};
void build_mel_runs(const mm_config& c, const MelSweep& sw, int n_waves, MelRuns* out);

int validate(const mm_config* c);
void build_window(const mm_config& c, float* out);
void build_mel(const mm_config& c, float* out);
void build_dct(const mm_config& c, float* out);
void build_twiddles(int n, float* out);
int build_butter_sos(int order, double wn, double* sos);
void build_mel_csr(const mm_config& c, const float* dense, MelCsr* csr);

}  // namespace mm
#endif
