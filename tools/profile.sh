#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# passes (MI355X_MICROARCH.md: the two do not fit one pass).  Outputs under gpurun_out/$1/.
set -o pipefail
tag=${1:-prof}; shift
# remaining args go to bench.py (e.g. --variant m12, --workload c4).  The trace pass runs the bench WITH its other
# sections and 50 warm-up + 100 timed steps: the tracked kernel average must be a steady-state one.
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 100 --warmup 50 --no-cpu --no-check "$@" > $out/bench_trace.json 2> $out/trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-check --no-extra "$@" > $out/bench_fetch.json 2> $out/fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-check --no-extra "$@" > $out/bench_write.json 2> $out/write.log || exit 1
find $out -name "*.csv" | sort | sed -n 1,20p
