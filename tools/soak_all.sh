#!/bin/bash
# Randomised differential soak of every entry point, several processes side by side (the CPU oracle is the
# slow half).  Usage: tools/soak_all.sh TAG SECONDS FIRST_SEED
TAG=${1:-soak}; SECS=${2:-300}; SEED=${3:-2000000}
O=gpurun_out/${TAG}_soak; mkdir -p $O
i=0
for w in beam beam beam_long crf crf viterbi duplex duplex_long crf_duplex crf_greedy envelope; do
  i=$((i+1))
  timeout -k 10 $SECS python tools/soak.py 100000000 $((SEED + i*1000000)) $w > $O/$i.$w.log 2>&1 &
done
wait
# a soak killed by the time limit prints no summary: failures are whatever lines it printed
grep -H -i "mismatch\|kernel [0-9]*:\|Traceback" $O/*.log | head -40; tail -q -n1 $O/*.log
echo "soak logs: $(ls $O | wc -l), failing lines: $(cat $O/*.log | grep -c -i 'mismatch\|seed .* kernel\|Traceback')"
