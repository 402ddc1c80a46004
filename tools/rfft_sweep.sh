#!/bin/bash
# dev helper: build side copies of the library with 2/3/4 waves per SIMD for rfft16_kernel (never the
# product .so: the copies are loaded through MODMFCC_LIB) and print the stage-isolated figure
for w in 2 3 4; do
  (cd modulation_mfcc_amd/csrc && /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared --offload-arch=gfx950 -DMM_RFFT_WAVES_PER_SIMD=$w -x hip mm_unity.hip -x hip mm_tables.cpp -o ../libmodmfcc_rfft$w.so 2>/dev/null)
  for i in 1 2; do
  MODMFCC_LIB=$PWD/modulation_mfcc_amd/libmodmfcc_rfft$w.so python bench.py --no-cpu --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rfft_stage']; print('waves/SIMD', $w, 'rfft GB/s', round(r['achieved']), 'frac', round(r['frac'],3), 'ms', round(r['avg_launch_ms'],4), 'modspec', d['kernels_ms']['modspec'])"
  done
done
