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
# digests of the kernel sources the counters belong to (bench.py checks them before quoting the numbers)
python -c "import json, bench; json.dump(bench.kernel_source_digest(), open('$OUT/kernel_source_md5.json', 'w'))"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $R/tools/prof_workload.py $WHICH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python $R/tools/prof_workload.py $WHICH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python $R/tools/prof_workload.py $WHICH > $OUT/write.log 2>&1
ls -la $OUT
