"""`python bench.py --gpus N` must start by itself: with WORLD_SIZE unset it becomes the launcher of
N ranks of itself.  Here on the CPU: the launcher, the rendezvous, the barriers and the max-over-ranks clock with a
stand-in for the step (--stub: gloo, no search); and the refusal when the node has fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_2_launches_two_ranks_by_itself():
    r = run("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    per = d["ranks"]["per_rank_ms_per_step"]
    assert len(per) == 2 and abs(d["ms_per_step"] - max(per)) < 1e-6  # the slowest rank's clock
    assert abs(d["value"] - 2 * d["config"]["reads_per_gpu"] * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]


def test_bench_config3_goes_the_same_way():
    r = run("--gpus", "2", "--steps", "2", "--warmup", "0", "--stub", "--config", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["baseline_config"] == 3 and d["config"]["reads_per_gpu"] == 8192


def test_bench_refuses_more_gpus_than_the_node_has():
    """(this container has none: the real path must say so and fail, not run something smaller)"""
    r = run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode == 2 and "GPU(s)" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_viterbi_clock_note_reads_the_committed_summary():
    """bench.py quotes both clocks of the viterbi kernel from profiles/*_viterbi_clock_summary.json (a helper that runs
    inside the timed script on the GPU box: it must not be able to take the bench line down)"""
    import bench
    note = bench.viterbi_clock_note()
    assert note is not None and "error" not in note, note
    assert 0.9 < note["events_over_rocprofv3"] < 1.1
    assert os.path.exists(os.path.join(bench.ROOT, note["source"]))
