#!/bin/bash
# dev helper: build libmodmfcc.so and print the resource usage of the kernels whose mangled name matches $1
cd "$(dirname "$0")/../modulation_mfcc_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wall -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -x hip mm_kernels.hip -x hip mm_tables.cpp -o ../libmodmfcc.so 2> /tmp/mm_build.log
rc=$?
grep -E "error|warning: " /tmp/mm_build.log | head -20
[ -n "$1" ] && grep -E "Function Name: .*$1" -A9 /tmp/mm_build.log | grep -E "Function Name|VGPRs:|Scratch|Spill|SGPRs:" | sed 's/.*remark: *//'
exit $rc
