#!/usr/bin/env python3
"""Benchmark of the hot path: projected ranges / second on BASELINE.json's
headline workload (1M-record synthetic PAF, 100k-range BED, `-x -m 3`).

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (lookup + CIGAR projection + transitive
frontier updates) over the whole query batch, index and query ranges already
resident in HBM.  Prints ONE JSON line (rank 0).  See DESIGN.md section 7.
"""
import argparse
import json
import glob
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_PROJECTION = 856  # 32 B entry + 4 B x 200 ops + 24 B result
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


T_START = time.time()


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


def flush_c_stdio():
    """Flush libc's stdio buffers (libraries such as RCCL print through them; when stdout is a pipe their
    text would otherwise appear at process exit, after the JSON line)."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=1_000_000)
    ap.add_argument("--ranges", type=int, default=100_000, help="query ranges per GPU")
    ap.add_argument("--max-depth", type=int, default=3)
    ap.add_argument("--no-transitive", action="store_true")
    ap.add_argument("--chunk-ranges", type=int, default=50000)
    ap.add_argument("--pair-budget", type=int, default=1 << 30)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="ranges timed on the CPU oracle, ~15 s of CPU work (0 = skip)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="KEY=VALUE",
                    help="impg_gpu_set_option before the run (timing comparisons, e.g. locality_min=0)")
    ap.add_argument("--paf", default=None, help="reuse an existing synthetic PAF file")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path even with one rank")
    args = ap.parse_args()

    # stdout carries exactly one line, the result: anything a library prints to fd 1 (RCCL's version
    # banner does) is sent to stderr instead
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch

    import impg_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- synthetic inputs (SplitMix64 seeds 42 / 7) -------
    n_seq, seq_len = 200, 5_000_000
    paf = args.paf or os.path.join(tempfile.gettempdir(), "impg_synth_%d_seed42.paf" % args.records)
    if rank == 0 and not os.path.exists(paf):
        log("writing synthetic PAF %s" % paf)
        impg_amd.synth_paf_text(paf + ".tmp", 42, args.records, n_seq=n_seq, seq_len=seq_len)
        os.replace(paf + ".tmp", paf)
    if dist is not None:
        dist.barrier()
        flush_c_stdio()  # (the communicator exists now: its banner, if any, goes out first)
    transitive = not args.no_transitive
    params = impg_amd.make_params(transitive=transitive, max_depth=args.max_depth)  # -x -m 3, defaults otherwise

    log("building the device index")
    t_build = time.time()
    if dist is None:
        index = impg_amd.GpuImpg.from_paf(paf, device=local_rank)
        engine = None
    else:
        from impg_amd.sharded import ShardedImpg
        engine = ShardedImpg.from_paf(paf, rank, world, device=local_rank)
        engine.chunk_ranges = args.chunk_ranges
        index = engine.local
    t_build = time.time() - t_build
    index.set_option("chunk_ranges", args.chunk_ranges)
    index.set_option("pair_budget", args.pair_budget)
    for kv in args.engine_option:
        k, v = kv.split("=", 1)
        index.set_option(k, int(v))

    # each rank is home to its own `--ranges` queries (weak scaling): seed 7 + rank
    bed = impg_amd.synth_bed(7 + rank, args.ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
    name_to_id = {impg_amd.synth_seq_name(k): index.seq_id(impg_amd.synth_seq_name(k)) for k in range(n_seq)}
    ranges = np.zeros(args.ranges, dtype=impg_amd.RANGE_DTYPE)
    ranges["target_id"] = [name_to_id[impg_amd.synth_seq_name(int(t))] for t in bed["target_id"]]
    ranges["start"], ranges["end"] = bed["start"], bed["end"]
    d_ranges = torch.from_numpy(ranges.view(np.uint8)).to(dev)  # resident in HBM before the timed region

    def step():
        if engine is None:
            st, _, _ = index.query_batch_stats(None, params, counts=False, checksums=False,
                                               device_ptr=d_ranges.data_ptr(), n=args.ranges)
            return st
        return engine.query_batch_stats(d_ranges, args.ranges, params)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    log("index ready (%.1f s, %.2f GB in HBM); warmup" % (t_build, index.device_bytes() / 1e9))
    for _ in range(args.warmup):
        st = step()
        log("warmup step: %d projected, engine %.1f ms (lookup %.1f project %.1f update %.1f)" %
            (st.projected, st.ms_total, st.ms_lookup, st.ms_project, st.ms_update))
    sync()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0
    log("timed region: %.3f s for %d steps" % (dt, args.steps))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(sum(s.projected for s in stats))], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        projected_total = float(tot.item())
    else:
        projected_total = float(sum(s.projected for s in stats))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        flush_c_stdio()
        return

    proj_per_step = sum(s.projected for s in stats) / max(1, args.steps)
    ms_project = sum(s.ms_project for s in stats)
    launches = sum(s.project_launches for s in stats)
    ach = (sum(s.projected for s in stats) * ALG_BYTES_PER_PROJECTION) / (ms_project * 1e-3) / 1e9 if ms_project > 0 else 0.0
    # HBM bytes per projection measured with rocprofv3 PMC passes of this same command
    # (scripts/profile_r1.sh -> profiles/r1_v2_traffic.json; FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
    traffic = None
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))  # newest build last
    tpath = tfiles[-1] if tfiles else ""
    if os.path.exists(tpath) and launches:
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj["hbm_bytes_per_pair"] * (sum(s.pairs for s in stats) or sum(s.projected for s in stats)) / launches
    out = {
        "metric": "projected ranges/sec, 1M-PAF 100k-BED -x depth 3; CPU coitrees baseline",
        "value": projected_total / dt,
        "unit": "projected ranges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / max(1, args.steps),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": "synthetic PAF %d records (200 seqs x 5 Mb, 10 kb alignments, 200-op CIGARs, bidirectional "
                        "index), %d x 5 kb query ranges per GPU, %s" %
                        (args.records, args.ranges, ("-x -m %d" % args.max_depth) if transitive else "no transitive"),
            "records": args.records, "ranges_per_gpu": args.ranges, "max_depth": args.max_depth if transitive else 0,
            "min_transitive_len": 101, "min_distance_between_ranges": 10,
            "parallelism": "1 process/GPU, index sharded by target sequence" if world > 1 else "single GPU",
            "chunk_ranges": args.chunk_ranges, "pair_budget": args.pair_budget,
        },
        "projected_per_step_rank0": proj_per_step,
        "pairs_per_step_rank0": sum(s.pairs for s in stats) / max(1, args.steps),
        "frontier_ranges_per_step_rank0": sum(s.frontier_ranges for s in stats) / max(1, args.steps),
        "stage_ms_per_step_rank0": {"lookup": sum(s.ms_lookup for s in stats) / max(1, args.steps),
                                    "project": ms_project / max(1, args.steps),
                                    "update": sum(s.ms_update for s in stats) / max(1, args.steps),
                                    "engine_total": sum(s.ms_total for s in stats) / max(1, args.steps)},
        "index_build_s": t_build,
        "index_bytes": index.device_bytes(),
        "roofline": {
            "bound": "hbm", "kernel": "project_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_note": "bytes per launch = rocprofv3 (FETCH_SIZE x2 [gfx950] + WRITE_SIZE) per pair, from "
                            "profiles/%s, x pairs per launch of this run" % os.path.basename(tpath),
            "achieved_traffic_GBs": (traffic / (ms_project / launches * 1e-3) / 1e9) if (traffic and ms_project) else None,
            "algorithmic_bytes_per_projection": ALG_BYTES_PER_PROJECTION,
            "launches": launches,
            "avg_launch_ms": ms_project / launches if launches else None,
            "avg_projections_per_launch": (sum(s.projected for s in stats) / launches) if launches else None,
        },
    }
    out["cpu_baseline"] = cpu_baseline(args, paf, ranges, transitive) if (world == 1 and args.cpu_sample > 0) else None
    if dist is not None:
        dist.destroy_process_group()
    flush_c_stdio()
    result_out.write(json.dumps(out) + "\n")
    result_out.flush()


def cpu_baseline(args, paf, ranges, transitive):
    """The C++ restatement of the reference's CPU path (oracle, kind "port"),
    timed on this box's host cores on a bounded sample of the same workload:
    ranges serial, each BFS level's frontier parallel over all cores, CIGARs
    re-read with pread + ASCII parse per hit -- the reference's structure
    (src/main.rs:7435, src/impg.rs:2384-2465, :495-551)."""
    from oracle import oracle as o
    cores = os.cpu_count() or 1
    log("cpu baseline: building the oracle index")
    t0 = time.time()
    ix = o.OracleIndex(paf_paths=[paf], preparse=False)
    build_s = time.time() - t0
    log("cpu baseline: oracle index built in %.1f s; timing the sample" % build_s)
    n = min(args.cpu_sample, len(ranges))
    p = o.make_params(transitive=transitive, max_depth=args.max_depth)
    sub = ranges[:n]
    proj, nres, sec = ix.bench(sub["target_id"], sub["start"], sub["end"], p, threads=cores, mode=0)
    return {"value": proj / sec if sec > 0 else 0.0, "unit": "projected ranges/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d query ranges, same PAF and flags (%d projections in %.2f s); ranges serial, "
                      "frontier parallel over %d threads, per-hit pread + CIGAR parse" % (n, len(ranges), proj, sec, cores),
            "oracle_index_build_s": build_s}


if __name__ == "__main__":
    main()
