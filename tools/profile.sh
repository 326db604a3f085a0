#!/bin/bash
# rocprofv3 recipe for the GPU box: kernel trace + stats, then HBM byte counters in their own
# passes (the pool refuses --pmc combined with other trace domains).  Usage: tools/profile.sh TAG [which]
set -u
TAG=${1:-r01}
WHICH=${2:-all}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $R/tools/prof_workload.py $WHICH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python $R/tools/prof_workload.py $WHICH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python $R/tools/prof_workload.py $WHICH > $OUT/write.log 2>&1
ls -la $OUT
