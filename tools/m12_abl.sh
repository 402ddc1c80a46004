#!/bin/bash
# dev: build ablated copies of the library into /tmp-like side files (never the product .so) and time them
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/abl
for m in "$@"; do
  (cd modulation_mfcc_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DMM_M12_ABL=$m \
     -x hip mm_kernels.hip -x hip mm_tables.cpp -o ../libmodmfcc_abl$m.so 2>/dev/null) &
done
wait
ls -la modulation_mfcc_amd/libmodmfcc_abl*.so
