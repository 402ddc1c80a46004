#!/bin/bash
# dev helper: per-kernel register / spill report of the kernels whose mangled name matches $1.  Builds with the
# Makefile's own flags (`make ru`) into modulation_mfcc_amd/libmodmfcc_ru.so -- the product library is not touched.
cd "$(dirname "$0")/../modulation_mfcc_amd/csrc" || exit 1
touch mm_unity.hip
make ru 2> /tmp/mm_build.log > /dev/null
rc=$?
grep -E "error|warning: " /tmp/mm_build.log | head -20
[ -n "$1" ] && grep -E "Function Name: .*$1" -A9 /tmp/mm_build.log | grep -E "Function Name|VGPRs:|Scratch|Spill|SGPRs:" | sed 's/.*remark: *//'
exit $rc
