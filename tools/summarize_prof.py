#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the small files kept under profiles/.

    python tools/summarize_prof.py gpurun_out/prof1 profiles/r01

Writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels + the largest
others, names shortened) and <prefix>_pmc.csv (FETCH_SIZE / WRITE_SIZE per launch, raw KB and
bytes corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950 -- calibrated here on
the stage-isolated rFFT kernel, whose read bytes are known exactly; WRITE_SIZE as is).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:60]


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    ours = ("logmel512", "logmel12m", "logmel_wpf", "stft_generic", "stft_any", "dct_clamp", "dct_fixup", "rfft16", "rfft_wpf", "rfft_generic", "mfcc_change", "decode_keys", "devcopy", "resample", "hb_pass", "hb_mid", "hb_row", "chg_", "pcm_decode", "rms_tile", "sos_seg", "clip_tab", "stencil")
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if any(o in r["Name"] for o in ours) or float(r["Percentage"]) > 1.0:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"]])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind in ("fetch", "write"):
        fs = glob.glob(os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if any(o in r["Kernel_Name"] for o in ours):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    bench = {}
    try:
        bench = json.load(open(os.path.join(src, "bench_trace.json")))
    except Exception:
        pass
    # the kernel sources the counters were taken on: bench.py stamps its JSON with a content hash of csrc/ (computed on
    # the GPU box, from the snapshot that ran); bench.py's pmc_traffic() only trusts a summary whose stamp equals the
    # hash of the sources it runs on
    shas = set()
    for jf in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        try:
            shas.add(json.load(open(os.path.join(src, jf)))["config"]["csrc_sha16"])
        except Exception:
            pass
    sha = shas.pop() if len(shas) == 1 else "unknown"
    with open(prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg",
                    "read_bytes_corrected(x2)", "write_bytes", "hbm_bytes_per_launch", "csrc_sha16"])
        for k, v in sorted(pmc.items()):
            fe = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0])))
            wr = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
            w.writerow([k, len(v.get("FETCH_SIZE", [])), f"{fe:.1f}", f"{wr:.1f}",
                        int(2 * fe * 1024), int(wr * 1024), int(2 * fe * 1024 + wr * 1024), sha])
    if bench:
        with open(prefix + "_bench.json", "w") as f:
            json.dump(bench, f, indent=1)
    print(open(prefix + "_kernel_stats.csv").read())
    print(open(prefix + "_pmc.csv").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
