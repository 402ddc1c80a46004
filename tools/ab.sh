#!/bin/bash
# dev helper (GPU box): fused-kernel time of the n_fft = 512 variants on the same box, interleaved
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["kernel_path"], d["kernels_ms"]["logmel"], "ms_per_step", round(d["ms_per_step"],4))'
for i in 1 2 3; do
  for mp in 0 2; do
    MM_PATH=$mp python bench.py --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "$P"
  done
done
