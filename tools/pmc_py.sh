#!/bin/bash
# usage: tools/pmc_py.sh <tag> "<counter list>" <python script> [args]   (runs on the GPU box)
# One rocprofv3 --pmc pass over a python tool (the program itself right behind `--`), counters averaged per kernel.
tag=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 "$@" > $out/stdout.txt 2> $out/log.txt || { tail -5 $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "at::" in n or "rocclr" in n: continue
    agg[n.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
