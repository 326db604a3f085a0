"""Summarise a tools/profile_sq.sh run (gpurun_out/prof_TAG/sq{1,2}_counter_collection.csv) into
profiles/TAG_sq_counters.json: SQ counters per launch and per wavefront-step of the kernel named.

    python tools/summarize_sq.py TAG [kernel-substring] [reads] [T] [reads_per_wave]
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "beam_wave_kernel<5, 6, 2, 0"
reads = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
rpw = int(sys.argv[5]) if len(sys.argv) > 5 else 2
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
vals = collections.defaultdict(list)
name = None
for fn in ("sq1_counter_collection.csv", "sq2_counter_collection.csv"):
    for row in csv.DictReader(open(os.path.join(base, fn))):
        if pat in row["Kernel_Name"]:
            name = row["Kernel_Name"]
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
per_launch = {k: sum(v) / len(v) for k, v in vals.items()}
wave_steps = reads / rpw * T
out = {
    "kernel": name, "workload": "%d reads T=%d N=5 beam 5 thr 0.1 (BASELINE config 2), %d launches averaged"
    % (reads, T, len(next(iter(vals.values())))),
    "command": "tools/profile_sq.sh (rocprofv3 --kernel-trace --pmc <8 SQ counters>, two passes)",
    "counters_per_launch": per_launch,
    "per_wave_step": {k: v / wave_steps for k, v in per_launch.items()},
    "kernel_source_md5": json.load(open(os.path.join(base, "kernel_source_md5.json"))),
    "notes": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = shader cycles)",
}
dst = os.path.join(ROOT, "profiles", tag + "_sq_counters.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["per_wave_step"], indent=1))
