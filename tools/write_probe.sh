#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the beam kernels on the profile workload (one PMC pass each).  Usage: tools/write_probe.sh [which]
export TMPDIR=/tmp
R=$PWD
W=${1:-beam}
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/wp_$c -o w -- python $R/tools/prof_workload.py $W > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/wp_$c/**/w_counter_collection.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]; i=n.find("beam_")
    if i>=0: d[n[i:i+44]].append(float(r["Counter_Value"]))
for k,v in d.items():
    if "beam" in k: print("$c", k, "GB per launch %.3f" % (sorted(v)[len(v)//2]*1024/1e9))
PY
done
