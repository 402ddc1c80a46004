#!/bin/bash
# dev helper: run bench.py under several MM_DEBUG ablation masks and print per-kernel times
for d in "$@"; do
  MM_DEBUG=$d timeout -k 10 100 python bench.py --no-cpu --steps 10 2>/dev/null > /tmp/ab_$d.json
  python - "$d" <<'PY'
import json, sys
d = json.load(open(f"/tmp/ab_{sys.argv[1]}.json"))
print("dbg", sys.argv[1], d["kernels_ms"], "value %.3g" % d["value"])
PY
done
