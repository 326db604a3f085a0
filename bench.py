#!/usr/bin/env python
"""bench.py -- the driver's benchmark contract for the MI355X-native CTC decoder.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- search::beam_search (beam_size 5, beam_cut_threshold 0.1,
collapse_repeats) over one batch of 4096 synthetic reads of T=4000 x N=5 f32 posteriors per GPU
(BASELINE.json configs[1]) -- with the posteriors already resident in HBM when the timed region
starts.  Reads are independent, so N GPUs decode N independent shards (weak scaling, no collective
inside the search); for N > 1 each step ends with ONE RCCL gather of the decoded
(labels, path, lengths) to rank 0 over xGMI, inside the timed region.

Rank 0 prints ONE JSON line.  `value` is whole-job reads/s = N * batch * K / max-over-ranks time.
`roofline` prices the beam-search kernel against HBM bandwidth with ALGORITHMIC bytes
(T*N*4 in + 5 bytes per emitted label out) over the kernel's own duration measured
with HIP events on the launch stream.  `cpu_baseline` is the CPU oracle (a C restatement of the
reference's Rust, NOT the Rust itself) timed on this box's host cores on a bounded sample, whose
outputs are also compared with the GPU's.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, N, BEAM, THR = 4000, 5, 5, 0.1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)


def make_batch(seed, n_reads):
    """reference-style rows (tests/test_decode.py:15-17), seeded."""
    rng = np.random.default_rng(seed)
    x = rng.random((n_reads * T, N), dtype=np.float32)
    x /= np.linalg.norm(x, ord=2, axis=1, keepdims=True)
    return x.reshape(n_reads, T, N).astype(np.float32)


def make_batch_peaky(seed, n_reads):
    """SURVEY.md 8d's second, clearly labelled set: peaky softmax rows (logits ~ 4*N(0,1)), the shape
    real basecaller posteriors have -- most symbols fall under beam_cut_threshold.  NOT the metric's
    input; `--data peaky` exists to show how much of the reference-style figure is its worst-case density."""
    rng = np.random.default_rng(seed)
    z = 4.0 * rng.standard_normal((n_reads * T, N), dtype=np.float32)
    z -= z.max(axis=1, keepdims=True)
    np.exp(z, out=z)
    z /= z.sum(axis=1, keepdims=True)
    return z.reshape(n_reads, T, N).astype(np.float32)


def cpu_baseline(x_host, gpu_labels, gpu_path, gpu_len, budget_s):
    """Times the oracle (C restatement of src/search.rs -- NOT the Rust) on this box's host cores
    on a bounded sample of the same workload and checks the GPU's outputs against it.

    os.cpu_count() over-reports what a container may actually use, so the thread count is found
    empirically: short probes at 1, 2, 4, ... threads until throughput stops improving; the timed
    run uses the best count and ~budget_s seconds of wall time."""
    from oracle import oracle

    n_avail = x_host.shape[0]
    max_threads = os.cpu_count() or 1

    def run(n, threads, passes=1, out=None):
        out = out or oracle.batch_outputs(n, T)
        t0 = time.perf_counter()
        res = oracle.beam_search_batch(x_host[:n], BEAM, THR, True, threads, n_passes=passes, out=out)
        return n * passes / (time.perf_counter() - t0), res

    rate1, _ = run(32, 1)
    best_rate, best_threads = rate1, 1
    threads = 2
    while threads <= max_threads:
        n = min(n_avail, max(64, 4 * threads))
        rate, _ = run(n, threads)
        if rate > best_rate * 1.10:
            best_rate, best_threads = rate, threads
            threads *= 2
        else:
            break
    n = int(min(n_avail, max(64, best_rate * budget_s)))
    passes = int(max(1, min(256, best_rate * budget_s / n)))
    out = oracle.batch_outputs(n, T)  # pre-touched: page faults stay out of the timed call
    t0 = time.perf_counter()
    labels, path, lens, status = oracle.beam_search_batch(x_host[:n], BEAM, THR, True, best_threads,
                                                          n_passes=passes, out=out)
    dt = time.perf_counter() - t0
    mism = 0
    for i in range(n):
        L = int(lens[i])
        ok = status[i] == 0 and int(gpu_len[i]) == L \
            and np.array_equal(gpu_labels[i, :L], labels[i, :L]) \
            and np.array_equal(gpu_path[i, :L].astype(np.int64), path[i, :L])
        mism += 0 if ok else 1
    return {
        "value": n * passes / dt, "unit": "reads/s", "cores": best_threads, "kind": "port",
        "sample": "first %d reads of rank 0's batch x %d passes (T=%d N=%d beam=%d thr=%.1f), oracle C "
                  "restatement of src/search.rs, %d pthreads (best of a 1,2,4,.. probe; os.cpu_count()=%d), "
                  "%.1f s" % (n, passes, T, N, BEAM, THR, best_threads, max_threads, dt),
        "single_thread_reads_per_s": rate1,
        "gpu_vs_oracle_mismatches": mism, "gpu_vs_oracle_compared": n,
    }


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch from the newest committed PMC summary (profiles/*_pmc_summary.json,
    produced by tools/profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the
    same kernel at the same shape, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    for path in reversed(files):  # newest summary that holds this kernel
        with open(path) as f:
            summ = json.load(f)
        for name, e in summ.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and ", true, false" not in name and "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"], "%s: %s; %s; %s" % (
                    os.path.basename(path), name, e.get("workload", ""), e.get("fetch_correction_note", ""))
    return None, None


def valu_issue_roofline(kernel_prefix, kernel_ms, n_simd, clock_hz=2.4e9):
    """The bound the beam kernel actually runs against: VALU instruction issue.  achieved = wavefront-level
    VALU instructions per launch (SQ_INSTS_VALU from the newest committed profiles/*_sq_counters.json of
    this kernel at this shape: tools/profile_sq.sh, rocprofv3 --pmc in its own passes) / the kernel duration
    measured here; peak = one wave64 VALU instruction per SIMD every 2 cycles (SIMD-32,
    MI355X_MICROARCH.md 'v_fma_f32 (wave64): 2 cyc') x SIMDs x 2.4 GHz."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))):
        with open(path) as f:
            d = json.load(f)
        if kernel_prefix in (d.get("kernel") or "") and "SQ_INSTS_VALU" in d.get("counters_per_launch", {}):
            insts = d["counters_per_launch"]["SQ_INSTS_VALU"]
            per_step = d.get("per_wave_step", {})
            achieved = insts / (kernel_ms * 1e-3)
            peak = n_simd * clock_hz / 2.0
            return {
                "bound": "valu_issue", "achieved": achieved, "peak": peak, "unit": "wave64 VALU instructions/s",
                "frac": achieved / peak,
                "source": "%s: SQ_INSTS_VALU %.4g per launch (%s)" % (os.path.basename(path), insts, d.get("workload", "")),
                "per_wavefront_step": {k: round(per_step[k], 1) for k in
                                       ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
                                        "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in per_step},
                "note": "at 4096 reads only 2 wavefronts share a SIMD: the step time is the wavefront's own dependent "
                        "instruction chain (a lone wavefront per SIMD takes ~87 % as long per step), so the fraction "
                        "rises with the batch (profiles/r02*_cycle_account.jsonl, DESIGN.md 4.1)",
            }
    return None


def viterbi_roofline(fcd, torch, dev, n_reads=16384, reps=5):
    """The HBM-bound kernel of the path: viterbi_search on n_reads x T x N,
    timed with the C ABI's HIP events."""
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.rand((n_reads, T, N), generator=g, device=dev, dtype=torch.float32)
    x /= torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
    r = fcd.viterbi_search_batch_raw(x)
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(reps):
        r = fcd.viterbi_search_batch_raw(x)
    torch.cuda.synchronize()
    ms, calls = h.timing_mean_ms()
    mean_L = float(r.out_len.float().mean())
    bytes_per_read = T * N * 4 + 5.0 * mean_L
    achieved = n_reads * bytes_per_read / (ms * 1e-3) / 1e9
    traffic, note = pmc_traffic("viterbi_stream_kernel")
    return {
        "kernel": "viterbi_stream_kernel<5> (search::viterbi_search)", "bound": "hbm",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": note,
        "reads": n_reads, "kernel_ms": ms, "launches_timed": calls,
        "reads_per_s": n_reads / (ms * 1e-3), "algorithmic_bytes_per_read": bytes_per_read,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall budget of the CPU baseline leg")
    ap.add_argument("--kernel", type=int, default=0,
                    help="0 auto, 1 generic (LDS), 2 wave (two reads/wavefront), 3 wave (one read/wavefront)")
    ap.add_argument("--no-viterbi", action="store_true", help="skip the secondary viterbi roofline leg")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue successive steps round-robin on this many HIP streams (each with its own "
                         "handle and tree arena) so that independent batches overlap on the GPU; 1 = strictly "
                         "one batch after the other (the default, and what `value` is quoted on)")
    ap.add_argument("--data", choices=("reference", "peaky"), default="reference",
                    help="reference = the metric's generator (tests/test_decode.py:15-17 style rows); "
                         "peaky = softmax rows, a labelled secondary set")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: run each step's result gather on the compute stream instead of overlapping it "
                         "with the next step's search on a second HIP stream")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gather even with one rank (path check on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import fast_ctc_decode_amd as fcd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(backend="nccl", device_id=dev)

    B = args.batch
    x_host = (make_batch if args.data == "reference" else make_batch_peaky)(1 + rank, B)
    x = torch.from_numpy(x_host).to(dev)  # resident in HBM before the timed region
    torch.cuda.synchronize()

    from fast_ctc_decode_amd import dist as fdist
    from fast_ctc_decode_amd import _native as nat
    counts = [B] * world
    scratch = {}
    n_streams = max(1, args.streams)
    handles = [nat.default_handle(local_rank)] + [nat.Handle(local_rank) for _ in range(n_streams - 1)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
    step_no = [0]
    # The gather of step i (RCCL over xGMI, the out_len-prefixed payload packed on the GPU: ~6 KB per read into
    # rank 0) runs on its own HIP stream and overlaps the search of step i+1.  Sizing the payload reads two
    # small values back to the host, so the host issues it one step LATE -- after the search of step i+1 has
    # been queued -- and never leaves the GPU without work; it only reads step i's result tensors, which every
    # call allocates afresh.  flush() issues the last one; all gathers have completed before the closing synchronize.
    comm_stream = torch.cuda.Stream(dev) if distributed and not args.no_overlap else None
    pending = [None]

    def gather(prev):
        r, ev = prev
        comm_stream.wait_event(ev)
        with torch.cuda.stream(comm_stream):
            for tns in (r.labels, r.path, r.out_len, r.status):
                tns.record_stream(comm_stream)
            fdist.gather_batch_result(r, counts, dst=0, scratch=scratch)

    def flush():
        if pending[0] is not None:
            gather(pending[0])
            pending[0] = None

    def step():
        s = step_no[0] % n_streams
        step_no[0] += 1
        with torch.cuda.stream(streams[s]):
            r = fcd.beam_search_batch_raw(x, BEAM, THR, True, kernel=args.kernel, handle=handles[s])
            if distributed and comm_stream is None:
                # ONE gather of the packed results to rank 0 (RCCL over xGMI), on the compute stream
                fdist.gather_batch_result(r, counts, dst=0, scratch=scratch)
        if comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(streams[s])
            flush()  # the previous step's results, while this step's search runs
            pending[0] = (r, ev)
        return r

    for _ in range(args.warmup):
        r = step()
    flush()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if args.warmup == 0:
        step()
        flush()
    torch.cuda.synchronize()
    for hh in handles:
        hh.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    flush()  # K searches and K gathers inside the timed region
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # kernel duration: the C ABI brackets every launch of the timed region with a HIP event pair
    # on the launch stream (torch's current stream); read them back after the final sync.
    tm = [hh.timing_mean_ms() for hh in handles]
    k_calls = sum(n for _, n in tm)
    k_ms = sum(ms * n for ms, n in tm) / max(k_calls, 1)

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())

    if rank == 0:
        rc = r.cpu()
        ok = int((rc.status == 0).sum())
        mean_L = float(rc.out_len.astype(np.float64).mean())
        bytes_per_read = T * N * 4 + 5.0 * mean_L  # posteriors in, u8 label + u32 time out
        achieved = B * bytes_per_read / (k_ms * 1e-3) / 1e9
        # the CPU leg (and its output cross-check) runs on rank 0 at N = 1 only, as the contract asks
        cpu = cpu_baseline(x_host, rc.labels, rc.path, rc.out_len, args.cpu_seconds) if world == 1 else None
        # (the names carry further template arguments after S: counting / profiling / one-length flags)
        traffic, traffic_note = pmc_traffic("beam_wave_kernel<5, 6, 2, 0" if args.kernel in (0, 2)
                                            else "beam_wave_kernel<5, 8, 1, 0" if args.kernel == 3
                                            else "beam_generic_kernel")
        if args.batch != 4096 or args.data != "reference":
            traffic, traffic_note = None, None
        props = torch.cuda.get_device_properties(dev)
        simds = props.multi_processor_count * 4
        rpw = 2 if args.kernel in (0, 2) else 1
        # outside the timed region: the tie instrument on the same batch (include/fcd.h)
        amb = fcd.beam_search_batch_raw(x, BEAM, THR, True, count_ambiguous=True).cpu().ambiguous
        ties = {"reads_with_gt20_candidate_kept_tie": int((amb[:, 0] > 0).sum()),
                "reads_with_result_changing_tie": int((amb[:, 1] > 0).sum()),
                "reads_with_both": int(((amb[:, 0] > 0) & (amb[:, 1] > 0)).sum()),
                "note": "a read with either counter at 0 is pinned to the reference; the others are settled by the "
                        "oracle's exhaustive tie replay (tests/test_gpu_fullsize.py::test_config2_tie_instrument)"}
        vit = viterbi_roofline(fcd, torch, dev) if not args.no_viterbi else None
        valu = valu_issue_roofline("beam_wave_kernel<5, 6, 2, 0", k_ms, simds) \
            if (args.kernel in (0, 2) and args.batch == 4096 and args.data == "reference") else None
        out = {
            "metric": "reads/s (T=4000, N=5, beam=5)",
            "value": world * B * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "beam_search beam_size=5 beam_cut_threshold=0.1 collapse_repeats, "
                            "batch=4096 reads T=4000 N=5 per GPU (BASELINE.json configs[1]), "
                            + ("reference-style rows numpy default_rng(1+rank)" if args.data == "reference"
                               else "SECONDARY SET: peaky softmax rows (logits 4*N(0,1)), default_rng(1+rank)"),
                "reads_per_gpu": B, "T": T, "N": N, "beam_size": BEAM, "beam_cut_threshold": THR,
                "parallelism": ("reads sharded x%d, one RCCL gather of results per step%s"
                                % (world, "" if args.no_overlap else " (on a second stream, overlapping the next step)"))
                               if world > 1 else "single GPU",
                "kernel": {0: "auto (wave, two reads per wavefront)", 1: "generic-lds",
                           2: "wave-registers-2reads", 3: "wave-registers-1read"}[args.kernel],
                "reads_ok": ok, "mean_labels_per_read": mean_L, "streams": n_streams,
                "tie_instrument": ties,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "kernel": "beam search kernel, %.3f ms per launch (HIP events), %d reads x %.0f "
                          "algorithmic B/read" % (k_ms, B, bytes_per_read),
                "kernel_ms": k_ms, "launches_timed": k_calls,
                # The search is a serial chain of T dependent steps per read: it is bound by instruction
                # issue, not by HBM (SURVEY.md finding 5) -- priced here against the VALU issue peak.
                "secondary_bound": valu if valu is not None else {
                    "bound": "valu_issue", "achieved": None, "peak": simds * 2.4e9 / 2.0,
                    "note": "no SQ counter summary of this kernel / shape under profiles/"},
                "wavefronts_per_simd": B / rpw / simds,
                "step_latency_us": k_ms * 1e3 / T,
            },
            "cpu_baseline": cpu,
            "viterbi_roofline": vit,
        }
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
