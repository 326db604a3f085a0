#!/bin/bash
# developer scratch: dynamic instruction counts of the quicksort probe kernel (per wavefront = per list)
set -u
R=$PWD; O=$R/gpurun_out/prof_sortsq; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for tag in ${SORTSQ_CFGS:-25_1_32 64_3_32 130_3_32}; do  # n_planes_keep
  cfg=$(echo $tag | tr '_' ' ')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O -o s_$tag -- python $R/tools/dev/probe_sort_workload.py $cfg > $O/s_$tag.log 2>&1
  python - $O $tag <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
f = glob.glob("%s/**/s_%s_counter_collection.csv" % (O, tag), recursive=True)
if not f:
    print(tag, "no counter file"); sys.exit(0)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "wave_probe" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
w = acc.get("SQ_WAVES", 1) or 1
print(tag, "launches", n.get("SQ_WAVES"), "per wavefront:", {k: round(v / w, 1) for k, v in sorted(acc.items())})
PY
done
