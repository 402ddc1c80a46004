#!/bin/bash
# dev helper (GPU box): rebuild the library with -DMM_ABL=<mask> and print the fused kernel's time.
# bits: 1 constants 2-of-8 LDS reads, 2 no 16x16 exchange, 4 no ds_bpermute partners, 8 one power write
# per lane instead of 17 (w16 only), 16 no sample loads / LDS sample reads, 32 no mel phase, 64 no staging
# (32 / 64: w16s only).  Extra args after "--" go to the environment, e.g. MM_PATH=4.  Results are WRONG by construction; timing only.
set -e
for m in "$@"; do
  (cd modulation_mfcc_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DMM_ABL=$m -x hip mm_kernels.hip -x hip mm_tables.cpp -o ../libmodmfcc.so 2>&1 | grep -E "error" || true)
  for i in 1 2; do
  python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MM_ABL', $m, d['kernels_ms'])"
  done
done
