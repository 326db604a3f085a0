#!/bin/bash
# SQ issue / stall / instruction-fetch counters of the duplex kernels (PMC passes on their own, csv).
# Usage: tools/profile_duplex_sq.sh TAG
set -u
TAG=${1:-dsq}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT -o sq1 -- python $R/tools/prof_workload.py duplex > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT -o sq2 -- python $R/tools/prof_workload.py duplex > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o sq3 -- python $R/tools/prof_workload.py duplex > $OUT/sq3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_EXP_GDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o sq4 -- python $R/tools/prof_workload.py duplex > $OUT/sq4.log 2>&1
ls $OUT
tail -2 $OUT/sq1.log $OUT/sq2.log $OUT/sq3.log $OUT/sq4.log
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "duplex_kernel" not in k: continue
        acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(f.split("/")[-1], k)
        for c, x in sorted(v.items()): print("    %-28s %.4g" % (c, x))
PY
