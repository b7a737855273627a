#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on N GPUs of one node (one process per GPU, no collective on the
data path: the work units are independent).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--path dp|translate|both]

Prints ONE JSON line on rank 0.  `metric`/`value` are BASELINE.json's metric (profile-DP GCUPS, see
paramugsy_amd/dp.py); the measured numbers of the translate path -- the path the reference actually ships
 -- ride along in the same line under "translate".  Each carries
  roofline      achieved = algorithmic bytes per launch / average device time of the dominant kernel (HIP events
                on the launch stream), against the 8 TB/s HBM peak
  cpu_baseline  rank 0, N=1 only: the CPU side timed on this box's host cores on a bounded sample of the same
                workload (translate: the upstream reference binary oracle/_ref/m_translate when it travelled
                with the snapshot, kind "reference"; otherwise the oracle, kind "port")
Inputs are synthetic (seeded, paramugsy_amd/synth.py) and resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--path", choices=["dp", "translate", "both"], default="both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # translate workload (per rank): a Mugsy_profile node with 4+4 genomes of 1 Mbp
    ap.add_argument("--tr-genomes", type=int, default=4)
    ap.add_argument("--tr-genome-len", type=int, default=1000000)
    ap.add_argument("--tr-blocks", type=int, default=2500)
    ap.add_argument("--tr-deltas", type=int, default=16)
    ap.add_argument("--tr-entries", type=int, default=6000)
    # dp workload (per rank): BASELINE.json configs[1]
    ap.add_argument("--dp-pairs", type=int, default=10000)
    ap.add_argument("--dp-rows", type=int, default=2)
    ap.add_argument("--dp-len", type=int, default=1000)
    return ap.parse_args()


def dist_setup(n_gpus):
    """One process per GPU.  Returns (rank, world, torch, dist-or-None)."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libparamugsy_amd has no CPU path")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl")  # RCCL; used only for the timing barrier and the max-over-ranks
        dist = dist_mod
    assert world == n_gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    return rank, world, local, torch, dist


def timed_region(torch, dist, fn, steps, warmup):
    """W untimed steps, barrier + sync, exactly K steps, sync + barrier; returns the MAX wall time over ranks (s)."""
    for _ in range(warmup):
        fn()
    dev = torch.cuda.current_device()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(device_ids=[dev])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(device_ids=[dev])
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def measured_traffic(kernel_substr, config_key):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/pmc_traffic.json), if the bench
    configuration is the profiled one.  rocprofv3 cannot run inside bench.py; the file is written by
    tools/summarize_prof.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same script.
    FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes: MI355X_MICROARCH.md, HBM); both
    counters are in KB."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    entry = table.get(config_key)
    if not entry:
        return None
    for name, c in entry.items():
        if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int((2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024)
    return None


def bench_translate(args, rank, world, local, torch, dist):
    from paramugsy_amd import synth
    from paramugsy_amd.translate import TranslateJob, Workload
    tmp = tempfile.mkdtemp(prefix="pm_bench_tr_%d_" % rank)
    # weak scaling: every rank translates its own shard (its own delta files against the node's two sides);
    # shards differ by seed, sizes are identical
    w = synth.make_workload(tmp, 20261003 + rank, n_left=args.tr_genomes, n_right=args.tr_genomes, genome_len=args.tr_genome_len,
                            n_blocks=args.tr_blocks, n_deltas=args.tr_deltas, entries_per_delta=args.tr_entries, mean_len=1500)
    wl = Workload.load(w.left_dir, w.right_dir, w.delta_paths)
    t = wl.tables()
    job = TranslateJob(t, device=local)
    stream = torch.cuda.current_stream().cuda_stream
    dt = timed_region(torch, dist, lambda: job.run(stream), args.steps, args.warmup)
    n_ent, n_off = job.sizes()
    # per-kernel device time with HIP events on the launch stream
    prof = [job.run_profiled(stream) for _ in range(max(3, min(args.steps, 10)))]
    ms_filter = sum(p[0] for p in prof) / len(prof)
    ms_count = sum(p[1] for p in prof) / len(prof)
    ms_scan = sum(p[2] for p in prof) / len(prof)
    ms_emit = sum(p[3] for p in prof) / len(prof)
    count_bytes, emit_bytes, n_live = job.kernel_bytes()
    units = t.n_units
    # the dominant kernel is whichever of the two unit passes is longer; both read the same tables, the emit
    # pass also writes the entries and offsets
    dom_ms = max(ms_count, ms_emit)
    bits = job.coordinate_bits()
    dom_name = "translate_kernel<%s, %s>" % ("true" if ms_emit >= ms_count else "false", "int" if bits == 32 else "long long")
    alg_bytes = emit_bytes if ms_emit >= ms_count else count_bytes  # algorithmic bytes of the dominant kernel's launch
    tr_key = "translate:%d:%d:%d:%d:%d" % (args.tr_genomes, args.tr_genome_len, args.tr_blocks, args.tr_deltas, args.tr_entries)
    out = {
        "metric": "translate work units/s (delta entry x left row x right row; m_translate.cc:625-647)",
        "value": units * world * args.steps / dt,
        "unit": "units/s",
        "ms_per_step": dt / args.steps * 1e3,
        "dtype": "int32" if bits == 32 else "int64",
        "config": {"workload": "Mugsy_profile node: %d+%d genomes x %d bp, %d blocks/side, %d delta files x %d entries per rank"
                   % (args.tr_genomes, args.tr_genomes, args.tr_genome_len, args.tr_blocks, args.tr_deltas, args.tr_entries),
                   "units_per_rank": units, "live_units": n_live, "entries_out": n_ent, "offsets_out": n_off,
                   "coordinate_bits": bits},
        "kernel_ms": {"filter+compact": ms_filter, "translate_kernel<count>": ms_count, "rocprim_scan_x2": ms_scan,
                      "translate_kernel<emit>": ms_emit},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(dom_name, tr_key),
                     "kernel": dom_name,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "latency/divergence-bound integer state machine, one lane per unit; not an HBM-bound kernel (SURVEY 8d)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
        ora = os.path.join(ROOT, "oracle", "_build", "oracle_m_translate")
        exe, kind = (ref, "reference") if os.path.exists(ref) else (ora, "port")
        if os.path.exists(exe):
            t0 = time.perf_counter()
            r = subprocess.run([exe, w.left_dir, w.right_dir, w.list_path, os.path.join(tmp, "cpu.delta")])
            cpu_dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": units / cpu_dt, "unit": "units/s", "cores": 1, "kind": kind,
                                   "sample": "the whole per-rank job through the CLI (parse + translate + write), %d units, %.2f s, rc %d"
                                   % (units, cpu_dt, r.returncode)}
            # whole-job time of the drop-in CLI on the same files, PCIe and text I/O included
            cli = os.path.join(ROOT, "bin", "m_translate")
            if os.path.exists(cli):
                t0 = time.perf_counter()
                r2 = subprocess.run([cli, w.left_dir, w.right_dir, w.list_path, os.path.join(tmp, "gpu.delta")])
                cli_dt = time.perf_counter() - t0
                same = open(os.path.join(tmp, "gpu.delta"), "rb").read() == open(os.path.join(tmp, "cpu.delta"), "rb").read()
                out["cli_whole_job"] = {"units_per_s": units / cli_dt, "seconds": cli_dt, "rc": r2.returncode,
                                        "bytes_identical_to_cpu_baseline": bool(same)}
    job.close()
    wl.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def bench_dp(args, rank, world, local, torch, dist):
    """One step = one pass (fill + traceback) over this rank's profile pairs, inputs resident in HBM."""
    import numpy as np
    from paramugsy_amd import dp as dpm
    rows, L, n = args.dp_rows, args.dp_len, args.dp_pairs
    inputs = dpm.synth_pairs_fast(20261003 + rank, n, rows, L)
    params = dpm.make_params(rows, rows)
    batch = dpm.DpBatch(inputs, params, device=local)
    stream = torch.cuda.current_stream().cuda_stream
    dt = timed_region(torch, dist, lambda: batch.run(True, stream), args.steps, args.warmup)
    info = batch.info()
    prof = [batch.run_profiled(True, stream) for _ in range(max(3, min(args.steps, 10)))]
    ms_fill = sum(p[0] for p in prof) / len(prof)
    ms_tb = sum(p[1] for p in prof) / len(prof)
    cells = info["cells"]
    # algorithmic bytes per launch: 0.5 byte of decisions per cell + the packed columns read
    # (B once, A once per stripe) + the scores; the skew padding of the decision buffer is not counted
    alg_bytes = cells // 2 + info["input_bytes"] + n * 4
    # int32 VALU peak: every non-packed int32 VALU instruction measured at 4 cycles per wave64 instruction per
    # SIMD (tools/ubench/valu_rates.hip, profiles/r01_valu_rates.txt) -> 16 lanes/clk x 4 SIMDs x 256 CUs x 2.4 GHz
    valu_peak = 256 * 4 * 16 * 2.4e9
    variant = batch.variant()
    ops_per_cell = variant["valu_ops_per_cell"]  # csrc/dp_kernels.hip: column score 2 or 3, E 3, F 3, H + flags 5, H - open 1
    out = {
        "metric": "profile-DP GCUPS (global affine-gap profile x profile alignment, scores + traceback)",
        "value": cells * world * args.steps / dt / 1e9,
        "unit": "GCUPS",
        "ms_per_step": dt / args.steps * 1e3,
        "dtype": "int32",
        "config": {"workload": "%d synthetic %d-row x %d-column profile pairs per rank, int32 affine-gap scores, scores + full traceback"
                   % (n, rows, L), "pairs_per_rank": n, "rows": rows, "columns": L, "cells_per_step_per_rank": cells,
                   "chunks": info["chunks"], "kernel_variant": variant,
                   "reference_counterpart": "none: the reference has no DP (SURVEY.md 0); specification and oracle are this repo's own"},
        "kernel_ms": {"dp_fill_kernel": ms_fill, "dp_traceback_kernel": ms_tb},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (ms_fill * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (ms_fill * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": measured_traffic("dp_fill_kernel", "dp:%d:%d:%d" % (n, rows, L)), "kernel": "dp_fill_kernel",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "valu": {"ops_per_cell": ops_per_cell, "achieved_Tops": cells * ops_per_cell / (ms_fill * 1e-3) / 1e12,
                              "peak_Tops": valu_peak / 1e12, "frac": cells * ops_per_cell / (ms_fill * 1e-3) / valu_peak},
                     "note": "max-plus recurrence: the fill kernel is bound by int32 VALU issue, not by HBM; the HBM fraction is reported as measured"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # cpu_baseline leg: the oracle's scalar full-matrix aligner on a bounded sample of the same batch
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        k = max(1, min(n, int(2.0e9 // max(1, L * L))))  # about 2e9 cells
        sample = dpm.DpInputs(inputs.cols_a[:k * L], inputs.off_a[:k + 1], inputs.cols_b[:k * L], inputs.off_b[:k + 1])
        t0 = time.perf_counter()
        o_scores, o_paths = pyoracle.dp_align(sample, params)
        cpu_dt = time.perf_counter() - t0
        scores, ops, n_ops = batch.fetch()
        same = bool(np.array_equal(scores[:k], o_scores)) and all(
            np.array_equal(p, q) for p, q in zip(batch.paths(ops, n_ops)[:k], o_paths))
        out["cpu_baseline"] = {"value": k * L * L / cpu_dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                               "sample": "first %d pairs of the same batch through oracle/dp_oracle.c (scalar C, scores + paths), %.1f s; "
                                         "GPU results identical on the sample: %s" % (k, cpu_dt, same)}
    batch.close()
    return out


def main():
    args = parse_args()
    rank, world, local, torch, dist = dist_setup(args.gpus)
    result = {}
    tr = None
    if args.path in ("translate", "both"):
        tr = bench_translate(args, rank, world, local, torch, dist)
    dp = None
    if args.path in ("dp", "both"):
        dp = bench_dp(args, rank, world, local, torch, dist)
    main_part = dp if dp is not None else tr
    result.update(main_part)
    result.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "data": "synthetic"})
    if dp is not None and tr is not None:
        result["translate"] = tr
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
