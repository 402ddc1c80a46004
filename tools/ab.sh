#!/bin/bash
# dev helper (GPU box): same-box A/B of the n_fft = 512 kernel variants, interleaved, in ONE process per
# call (tools/m12_time.py pins variants through mm_plan_set_variant / mm_plan_set_fuse_dct; the library
# reads no environment variable).  usage: tools/ab.sh [variants...]   e.g. tools/ab.sh w16s w16s-nofuse m12
python tools/m12_time.py "${@:-w16s w16s-nofuse m12 w16s w16s-nofuse m12}"
