#!/bin/bash
# dev (GPU box): same-box A/B of library builds, interleaved, one process per measurement.
# usage: tools/ab_libs.sh <workload> <rounds> libA.so libB.so [...]     (paths relative to the repo root)
wl=$1; rounds=$2; shift 2
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    MODMFCC_LIB=$PWD/$lib timeout -k 10 120 python tools/time_workload.py $wl 200 || exit 1
  done
done
