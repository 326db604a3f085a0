#!/bin/bash
# tools/viterbi_clock.sh TAG -- the viterbi kernel timed by HIP events and by rocprofv3 on the SAME launches of the same
# process.  Writes gpurun_out/viterbi_clock_TAG/{events_plain.json, events_under_rocprof.json,
# trace_kernel_stats.csv, summary.json}.
set -u
TAG=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/viterbi_clock_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python $R/tools/viterbi_clock.py 100 f32 > $OUT/events_plain.json 2> $OUT/plain.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $R/tools/viterbi_clock.py 100 f32 > $OUT/events_under_rocprof.json 2> $OUT/rocprof.err
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
ev_plain = json.loads(open(os.path.join(out, "events_plain.json")).read().strip().splitlines()[-1])
ev_prof = json.loads(open(os.path.join(out, "events_under_rocprof.json")).read().strip().splitlines()[-1])
rows = []
for f in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "viterbi_stream_kernel" in r.get("Name", ""):
            rows.append({k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs") if k in r})
summary = {"hip_events_plain_process": ev_plain["kernel_ms_events"], "hip_events_under_rocprof": ev_prof["kernel_ms_events"],
           "rocprofv3_kernel_stats_same_process": rows, "clocks": {"before": ev_prof["clocks_before"], "after": ev_prof["clocks_after"]},
           "algorithmic_bytes_per_launch": ev_plain["algorithmic_bytes_per_launch"]}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary))
PY
