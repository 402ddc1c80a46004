// Single-translation-unit build of libmodmfcc (the `dev` and `ru` targets): in-kernel stamp arrays (-DMM_DEV) and the
// register report want every kernel in one compile.  The product library links the four units separately (Makefile).
#include "mm_api.hip"
#include "mm_logmel16s.hip"
#include "mm_reg2.hip"
#include "mm_tail.hip"
#include "mm_side.hip"
