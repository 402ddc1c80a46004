#!/bin/bash
# Runs on the GPU box (via gpurun): the GPU test suite, then the bench as the driver runs it, then the single-rank
# rehearsal of the N > 1 path.  A step that is killed by its timeout ends the call (no further GPU step after a hang).
# usage: tools/gpu_check.sh <tag> [pytest args]
tag=${1:-check}; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > $out/${tag}_tests.log 2>&1
rc=$?
tail -5 $out/${tag}_tests.log
if [ $rc -ge 124 ]; then echo "tests killed ($rc)"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rb=$?
if [ $rb -ge 124 ]; then echo "bench killed ($rb)"; exit $rb; fi
if [ $rb -ne 0 ]; then tail -20 $out/${tag}_bench.err; fi
MM_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-extra > $out/${tag}_dist.json 2> $out/${tag}_dist.err
rd=$?
if [ $rd -ne 0 ]; then tail -20 $out/${tag}_dist.err; fi
python - <<PY
import json
for f in ("$out/${tag}_bench.json", "$out/${tag}_dist.json"):
    try:
        r = json.load(open(f))
        print(f.split("/")[-1], "value %.4g" % r["value"], "ms/step %.4f" % r["ms_per_step"], "frac", r.get("roofline", {}).get("frac"),
              "parity_ok", r.get("parity_ok"), r.get("parity_checks"), "per_rank", r.get("per_rank"))
        for k in ("c2", "c4", "refdefault"):
            if k in r:
                print("  ", k, "ms/step %.4f" % r[k]["ms_per_step"], r[k].get("kernels_ms"), "frac", (r[k].get("roofline") or {}).get("frac"),
                      "ns/frame.mel %.4g" % r[k].get("ns_per_frame_mel", float("nan")), r[k].get("with_modspec", {}).get("ms_per_step"))
        if "refdefault_call" in r:
            print("  ", json.dumps(r["refdefault_call"])[:1500])
        if r.get("parity_ok") is False:
            raise SystemExit("parity check failed: %r" % r.get("parity_checks"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
rp=$?
exit $(( rc | rb | rd | rp ))
