#!/bin/bash
# Everything the per-round evidence under profiles/ is made from, in ONE call on the GPU box.
# Usage: tools/round_profiles.sh TAG      (then, here: tools/summarize_profile.py TAG; tools/summarize_sq.py TAG)
set -u
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
bash tools/profile.sh $TAG all > $O/${TAG}_profile.log 2>&1
bash tools/profile_sq.sh $TAG beam > $O/${TAG}_profile_sq.log 2>&1
# rocprofv3 summary of the bench command itself: its kernel average must agree with bench.py's HIP events
# (--no-e2e: the host-batch leg launches the same kernel on 1024-read chunks, which would blur the average)
( export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_bench -o bench -- \
    python $R/bench.py --no-e2e > $O/${TAG}_bench_line_rocprof.json 2> $O/${TAG}_bench_rocprof.err )
# (bench.py overlaps its steps and then times a few launches alone: the summary above averages both kinds -- the same
# command with every step in stream order, whose kernel average is a launch alone)
( export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_bench_overlap0 -o bench -- \
    python $R/bench.py --no-e2e --overlap 0 --cpu-seconds 1 > $O/${TAG}_bench_line_overlap0_rocprof.json 2> $O/${TAG}_bench_overlap0_rocprof.err )
# the default command, as the driver runs it (with the e2e leg)
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
# the BASELINE multi-GPU config's per-rank shard and the CRF config under the same contract
python bench.py --config 3 --no-viterbi > $O/${TAG}_bench_config3.json 2> $O/${TAG}_bench_config3.err
python bench.py --config 4 --no-viterbi > $O/${TAG}_bench_config4.json 2> $O/${TAG}_bench_config4.err
# BASELINE config 5 (the 2-D pair consensus) under the same contract: pairs/s, both log-add flavours
python bench.py --config 5 > $O/${TAG}_bench_config5.json 2> $O/${TAG}_bench_config5.err
python bench.py --config 5 --mode max > $O/${TAG}_bench_config5_max.json 2>> $O/${TAG}_bench_config5.err
python bench.py --config 5 --overlap 0 --steps 5 --warmup 1 --cpu-seconds 1 > $O/${TAG}_bench_config5_overlap0.json 2>> $O/${TAG}_bench_config5.err
python bench.py --config 5 --mode max --overlap 0 --steps 5 --warmup 1 --cpu-seconds 1 > $O/${TAG}_bench_config5_max_overlap0.json 2>> $O/${TAG}_bench_config5.err
# round 6: bench.py overlaps its steps on the handle's internal streams by default (fcd_set_overlap) -- the same lines with
# every step in stream order: one launch at a time, what the kernel-level roofline and the counters are quoted on
python bench.py --overlap 0 --no-e2e --cpu-seconds 1 > $O/${TAG}_bench_line_overlap0.json 2> $O/${TAG}_bench_overlap0.err
python bench.py --config 3 --overlap 0 --no-viterbi --steps 5 --cpu-seconds 1 > $O/${TAG}_bench_config3_overlap0.json 2>> $O/${TAG}_bench_overlap0.err
python bench.py --config 4 --overlap 0 --no-viterbi --cpu-seconds 1 > $O/${TAG}_bench_config4_overlap0.json 2>> $O/${TAG}_bench_overlap0.err
for n in 2 3 4 6 8; do python bench.py --config 3 --overlap $n --no-viterbi --steps 24 --warmup 8 --cpu-seconds 0.5 2>> $O/${TAG}_bench_config3_by_overlap.err; done > $O/${TAG}_bench_config3_by_overlap.txt
for n in 2 3 4 8; do python bench.py --overlap $n --no-viterbi --no-e2e --steps 40 --warmup 8 --cpu-seconds 0.5 2>> $O/${TAG}_bench_line_by_overlap.err; done > $O/${TAG}_bench_line_by_overlap.txt
python tools/duplex_overlap.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_duplex_overlap.txt
# the RCCL path of the bench at world size 1 (communicator, pack, ONE gather, one-launch unpack)
python bench.py --force-dist --no-viterbi --no-e2e --cpu-seconds 1 > $O/${TAG}_bench_rccl_world1.json 2> $O/${TAG}_bench_rccl_world1.err
python tools/probe_e2e.py 4096 --grid > $O/${TAG}_e2e_probe.txt 2>&1
python tools/duplex_account.py > $O/${TAG}_duplex_account.jsonl 2> $O/${TAG}_duplex_account.err
python tools/cycle_account.py > $O/${TAG}_cycle_account.jsonl 2> $O/${TAG}_cycle_account.err
python tools/bench_configs.py 1 3 4 5 64 1024 --check > $O/${TAG}_configs.jsonl 2> $O/${TAG}_configs.err
python tools/probe_latency.py > $O/${TAG}_latency.txt 2>&1
# what the tie order changes and costs, the viterbi kernel by both clocks, the bench variants
python tools/tie_order_delta.py 2 3 4 5 --oracle 8 > $O/${TAG}_tie_order_delta.jsonl 2> $O/${TAG}_tie_order_delta.err
bash tools/viterbi_clock.sh $TAG > $O/${TAG}_viterbi_clock.log 2>&1
cp $O/viterbi_clock_$TAG/summary.json $O/${TAG}_viterbi_clock_summary.json
cp $(find $O/viterbi_clock_$TAG -name '*kernel_stats.csv' | head -n 1) $O/${TAG}_viterbi_clock_kernel_stats.csv
FCD_TIE_ORDER=stable python bench.py --no-viterbi --no-e2e --cpu-seconds 1 > $O/${TAG}_bench_line_stable_order.json 2> $O/${TAG}_bench_stable.err
FCD_TIE_ORDER=stable python bench.py --config 3 --no-viterbi --cpu-seconds 1 > $O/${TAG}_bench_config3_stable_order.json 2>> $O/${TAG}_bench_stable.err
( python bench.py --batch 16384 --no-viterbi --no-e2e --cpu-seconds 1; python bench.py --data peaky --no-viterbi --no-e2e --cpu-seconds 1 ) > $O/${TAG}_bench_variants.txt 2> $O/${TAG}_bench_variants.err
# ---- round 5: the tie order's cost by beam (both orders, reference-style and peaky rows), its cycle account in place
# (a -DFCD_LANE_TIE_PROF build: tools/dev/lane_tie_prof.sh, made before the call), the replay's dynamic instruction
# counts, config 3 on 1 / 2 / 3 streams, the lane kernel's SQ counters, per-pair duplex callers through the coalescer
python tools/tie_order_by_beam.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_tie_order_by_beam.txt
python tools/dev/lane_tie_prof.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_lane_tie_prof.txt
python tools/dev/time_coop.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_replay_probe.txt
bash tools/dev/probe_sort_sq.sh 2>&1 | grep "per wavefront" > $O/${TAG}_replay_instruction_counts.txt
FCD_TIE_ORDER=stable bash tools/profile_sq.sh ${TAG}_lane_stable beam32 > $O/${TAG}_sq_lane_stable.log 2>&1
bash tools/profile_sq.sh ${TAG}_lane beam32 > $O/${TAG}_sq_lane.log 2>&1
python tools/probe_threads.py pairs 2000 3 1 16 64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_pair_callers.txt
python tools/probe_threads.py 4000 10 64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_read_callers.txt
for f in $O/${TAG}_bench_line.json $O/${TAG}_bench_config3.json $O/${TAG}_duplex_account.jsonl $O/${TAG}_e2e_probe.txt $O/${TAG}_cycle_account.jsonl $O/${TAG}_latency.txt; do tail -n 2 $f | cut -c1-300; done
