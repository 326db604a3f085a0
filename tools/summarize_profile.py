"""Summarise a tools/profile.sh run (gpurun_out/prof_TAG) into profiles/TAG_pmc_summary.json and
copy the kernel stats: python tools/summarize_profile.py TAG"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)

# workload of tools/prof_workload.py by kernel-name prefix (the names carry further template arguments)
WORK_PREFIX = [
    ("beam_wave_kernel<5, 6, 2, 0", "4096 reads T=4000 N=5 beam 5 thr 0.1 (BASELINE config 2)"),
    ("beam_wave_kernel<5, 6, 2, 4", "4096 reads T=4000 S=4 N=5 CRF beam 5 thr 0 (config 4)"),
    ("viterbi_stream_kernel<5", "16384 reads T=4000 N=5"),
    ("beam_generic_kernel", "8192 reads T=4000 N=5 beam 32 thr 0.1 (config 3, per GPU)"),
    ("beam_lane_kernel<5, 2", "8192 reads T=4000 N=5 beam 32 thr 0.1 (config 3, per GPU)"),
    ("duplex_kernel<0", "1024 pairs T=2000 band +-64 beam 5 thr 0.1 logsumexp (config 5), the any-shape kernel"),
    ("duplex_kernel<1", "1024 pairs T=2000 band +-64 beam 5 thr 0.1 max mode (config 5), the any-shape kernel"),
    ("duplex_slots_kernel<0", "1024 pairs T=2000 band +-64 beam 5 thr 0.1 logsumexp (config 5)"),
    ("duplex_slots_kernel<1", "1024 pairs T=2000 band +-64 beam 5 thr 0.1 max mode (config 5)"),
]


class _Work:
    def get(self, name, default=""):
        for prefix, text in WORK_PREFIX:
            if name.startswith(prefix):
                return text
        return default


WORK = _Work()


def short(n):
    m = re.search(r"(beam_wave_kernel<[^>]*>|beam_lane_kernel<[^>]*>|beam_generic_kernel|viterbi_stream_kernel<[^>]*>|viterbi_kernel|"
                  r"duplex_slots_kernel<[^>]*>|duplex_kernel<[^>]*>|crf_greedy_stream_kernel<[^>]*>|crf_greedy_kernel|envelope_kernel|ln_convert_kernel)", n)
    return m.group(1) if m else None


def agg(fn, counter):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(os.path.join(base, fn))):
        if row["Counter_Name"] == counter:
            n = short(row["Kernel_Name"])
            if n:
                d[n].append(float(row["Counter_Value"]))
    return d


f = agg("fetch_counter_collection.csv", "FETCH_SIZE")
w = agg("write_counter_collection.csv", "WRITE_SIZE")
stats = {}
for row in csv.DictReader(open(os.path.join(base, "trace_kernel_stats.csv"))):
    n = short(row["Name"])
    if n:
        stats[n] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                    "min_ms": float(row["MinNs"]) / 1e6, "max_ms": float(row["MaxNs"]) / 1e6}
out = {
    "source": "rocprofv3 on MI355X, tools/profile.sh %s (kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE "
              "in separate passes); workload tools/prof_workload.py" % tag,
    "units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch as reported; gfx950 FETCH_SIZE reads exactly 1/2 of the "
             "bytes of a wide coalesced (16 B/lane) stream (MI355X_MICROARCH.md HBM section) -- confirmed by "
             "viterbi_stream_kernel<5>: 16384 reads x 80000 B = 1 280 000 KiB of input vs its FETCH_SIZE",
    "kernels": {},
    # {file: md5} of the kernel sources this run was taken on (written next to the counters by tools/profile.sh on
    # the GPU box); bench.py quotes the numbers only while those files are unchanged
    "kernel_source_md5": json.load(open(os.path.join(base, "kernel_source_md5.json"))),
}
for k in stats:
    fs, ws = f.get(k, []), w.get(k, [])
    e = {"workload": WORK.get(k, ""), **stats[k], "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws}
    if fs and ws:
        fm, wm = sorted(fs)[len(fs) // 2], sorted(ws)[len(ws) // 2]
        wide = "viterbi_stream" in k
        e["fetch_correction"] = 2.0 if wide else 1.0
        e["fetch_correction_note"] = ("x2: 16 B/lane coalesced stream (calibrated above)" if wide else
                                      "x1: 4 B/lane accesses are uncalibrated on gfx950; the true value lies "
                                      "between x1 and x2")
        e["hbm_bytes_per_launch"] = (fm * e["fetch_correction"] + wm) * 1024
    out["kernels"][k] = e
dst = os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")
json.dump(out, open(dst, "w"), indent=1)
shutil.copy(os.path.join(base, "trace_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
cfg = os.path.join(base, "configs.jsonl")
if os.path.exists(cfg):
    shutil.copy(cfg, os.path.join(ROOT, "profiles", tag + "_configs.jsonl"))
for k, e in out["kernels"].items():
    print("%-34s avg %.3f ms  hbm bytes/launch %s" % (k, e["avg_ms"], e.get("hbm_bytes_per_launch")))
