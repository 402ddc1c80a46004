// What the any-length STFT kernels share across translation units: the parameter block (mm_anyfft.hip.inc in mm_api.hip,
// mm_reg2.hip) and the launch wrappers of the two-stage register kernels.
#pragma once
#include "mm_common.h"

struct __attribute__((packed, aligned(4))) MmAnyFloat2U { float x, y; };

struct AnyParams {
  const float* audio;
  int64_t batch, n_samples, stride, n_frames;
  int n_fft, hop, n_bins, n_mels;
  float preemph, amin, db_offset;
  const float* window;
  int nn, packed;               // complex transform length; 1 = even n_fft (packed real frame)
  int n_pass, radix[16];        // direct path: Stockham passes
  int M, log2M;                 // Bluestein path (M = 0: direct)
  const float2* tw;             // direct: exp(-2 pi i k / nn), k < nn; Bluestein: per stage, [half + pos] = W_(2 half)^pos (M entries)
  const float2* split;          // packed: exp(-2 pi i k / n_fft), k <= nn / 2
  const float2* chirp;          // Bluestein: w[j], j < nn
  const float2* bhat;           // Bluestein: FFT_M(b)[brev(p)] / M
  const int* mel_start;
  const int* mel_len;
  const int* mel_off;
  const float* mel_w;
  float* out_power;             // MODE 0: [B][T][n_bins]
  float* out_logmel;            // MODE 1: [B][n_mels][T]
  int* clip_key;                // MODE 1: [B]
  int frames_per_group;         // consecutive frames a thread group transforms
  unsigned grp_bytes;           // LDS bytes per thread group
  unsigned b_off, p_off;        // byte offsets of the second buffer and of the power row inside a group's LDS
  // LDSTAB: one packed copy of the constant tables (floats; ints as bits), copied to LDS behind the groups' buffers
  const float* tabpack;         // window | tw | split | chirp | mel_w | mel_start | mel_len | mel_off
  int tab_floats;
  int o_tw, o_split, o_chirp, o_melw, o_mstart, o_mlen, o_moff;     // float offsets inside the pack (window at 0)
};

// two register stages (mm_reg2.hip): nn = R1 x R2 complex points
bool reg2_pick(int nn, int* r1, int* r2);                       // is there an instantiation for this length?
size_t reg2_lds_bytes(int r1, int r2, int tab_floats);          // dynamic LDS of a launch (0: no such instantiation)
int reg2_frames_per_wave(int r1);                               // frames a wave walks (a multiple of its batch)
bool reg2_set_attr(int r1, int r2, int bytes);                  // hipFuncAttributeMaxDynamicSharedMemorySize (current device)
void reg2_launch(int r1, int r2, int mode, dim3 grid, size_t lds, hipStream_t st, const AnyParams& q);
