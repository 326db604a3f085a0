#!/bin/bash
# SQ issue/stall counters (PMC passes on their own, csv).  Usage: tools/profile_sq.sh TAG [beam|beam32|crf|viterbi]
set -u
TAG=${1:-sq}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# digests of the kernel sources the counters belong to (bench.py checks them before quoting the numbers)
python -c "import json, bench; json.dump(bench.kernel_source_digest(), open('$OUT/kernel_source_md5.json', 'w'))"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT -o sq1 -- python $R/tools/prof_workload.py ${2:-beam} > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT -o sq2 -- python $R/tools/prof_workload.py ${2:-beam} > $OUT/sq2.log 2>&1
ls $OUT
tail -3 $OUT/sq1.log $OUT/sq2.log
