// Internal declarations shared by mm_tables.cpp (host tables) and mm_kernels.hip (device + ABI).
#ifndef MM_INTERNAL_H
#define MM_INTERNAL_H

#include <vector>

#include "../../include/modmfcc.h"

namespace mm {

struct MelCsr {
  std::vector<int> start, len, off;  // per filter: first bin, run length, offset into w
  std::vector<float> w;
};

int validate(const mm_config* c);
void build_window(const mm_config& c, float* out);
void build_mel(const mm_config& c, float* out);
void build_dct(const mm_config& c, float* out);
void build_twiddles(int n, float* out);
int build_butter_sos(int order, double wn, double* sos);
void build_mel_csr(const mm_config& c, const float* dense, MelCsr* csr);

}  // namespace mm
#endif
