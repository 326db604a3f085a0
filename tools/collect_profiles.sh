#!/bin/bash
# After tools/round_profiles.sh TAG on the GPU box (gpurun_out/ is scratch): the summaries that are judged go to profiles/.
# Usage: tools/collect_profiles.sh TAG
set -u
TAG=${1:?tag}
O=gpurun_out; P=profiles
python tools/summarize_profile.py $TAG > /dev/null
python tools/summarize_sq.py $TAG > /dev/null 2>&1
python tools/summarize_sq.py ${TAG}_lane "beam_lane_kernel<5, 2" 8192 4000 2 > /dev/null 2>&1
python tools/summarize_sq.py ${TAG}_lane_stable "beam_lane_kernel<5, 2" 8192 4000 2 > /dev/null 2>&1
for f in $O/${TAG}_*.json $O/${TAG}_*.jsonl $O/${TAG}_*.txt; do
    case "$f" in *rocprof.err|*.log) continue;; esac
    [ -s "$f" ] && cp "$f" $P/
done
for d in bench bench_overlap0; do
    s=$(find $O/prof_${TAG}_$d -name '*kernel_stats.csv' 2>/dev/null | head -n 1)
    [ -n "$s" ] && cp "$s" $P/${TAG}_${d}_kernel_stats.csv
done
s=$(find $O/prof_$TAG -name 'trace_kernel_stats.csv' 2>/dev/null | head -n 1); [ -n "$s" ] && cp "$s" $P/${TAG}_kernel_stats.csv
grep -n "passed\|failed" $O/${TAG}_pytest_gpu.log | tail -n 1 > $P/${TAG}_pytest_gpu.txt
tail -n 1 $O/${TAG}_smoke.log >> $P/${TAG}_pytest_gpu.txt
bash tools/isa_table.sh > $P/${TAG}_isa_table.txt 2>/dev/null
ls $P | grep -c "^${TAG}_"
