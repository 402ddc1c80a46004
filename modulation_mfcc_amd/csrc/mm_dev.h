// Development instrumentation hooks of the fused kernels.  The PRODUCT build defines nothing here: every macro below
// expands to nothing and the kernels carry one-token markers only.  A side build with -DMM_DEV (tools/stamps*.py:
// `make -C modulation_mfcc_amd/csrc dev`, output ../libmodmfcc_dev.so, loaded through MODMFCC_LIB) pulls in
// mm_dev_stamps.inc: per-wave s_memtime totals of the loop sections of workgroup 0 (cdna_hip_programming.md section 7,
// "In-kernel stamps"), written to device arrays of their own that no kernel reads.
#pragma once
#ifdef MM_DEV
#include "mm_dev_stamps.inc"
#elif defined(MM_REGIONS)
// tools/isa_scan.py --mix: the stamp positions become assembly comments "; MMREG s<i>" that delimit the regions of the
// instruction-mix table (an ISA-only side compile, never a library)
#define MM_STAMP_BEGIN(N)
#define MM_STAMP_AT(i) asm volatile("; MMREG s" #i);
#define MM_STAMP_END(N)
#define MM_STAMP_END_BLK(N, B)
#define MM_STAMP_FWD_DECL
#define MM_STAMP_FWD(base)
#define MM_STAMP_REL(i) asm volatile("; MMREG r" #i);
#define MM_FIN_STAMP_BEGIN
#define MM_FIN_STAMP(i) asm volatile("; MMREG fin" #i);
#else
#define MM_STAMP_BEGIN(N)
#define MM_STAMP_AT(i)
#define MM_STAMP_END(N)
#define MM_STAMP_END_BLK(N, B)
#define MM_STAMP_FWD_DECL
#define MM_STAMP_FWD(base)
#define MM_STAMP_REL(i)
#define MM_FIN_STAMP_BEGIN
#define MM_FIN_STAMP(i)
#endif
