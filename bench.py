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
# Projections of one headline step (synth PAF seed 42, 1 000 000 records; BED seed 7, 100 000 ranges; -x -m 3, coitrees
# visit order).  Pinned by tests/test_gpu_fullsize.py::test_headline_timed_form, which runs the timed form beside the
# per-range counts (themselves checked against the oracle on a sample) -- the timed region asserts this very number.
HEADLINE_PROJECTED = 2_125_313_869
# Sum over the batch's 100 000 ranges of the per-range order-independent checksums (hit_stats_kernel's: a 64-bit mix of every
# row's query id, query interval, target id and target interval, added up mod 2^64).  Pinned by the same test, which ties the
# per-range checksums to the oracle on a sample; the self check recomputes it FROM THE ROWS the timed form leaves in HBM.
HEADLINE_CHECKSUM = 16_006_759_611_715_120_108  # (tests/test_gpu_fullsize.py::test_headline_timed_form asserts it)


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
    ap.add_argument("--records", type=int, default=None)
    ap.add_argument("--ranges", type=int, default=None, help="query ranges per GPU")
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--no-transitive", action="store_true")
    ap.add_argument("--chunk-ranges", type=int, default=None,
                    help="ranges per chunk (default: the whole batch on one GPU, 50000 per rank on a sharded index)")
    ap.add_argument("--pair-budget", type=int, default=None,
                    help="hit slots a level of a chunk may fill (default 3 x 2^30 on one GPU: the headline batch runs as one chunk, "
                         "85 GB of slot arrays on a 288 GB device; 2^30 on a sharded index)")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="ranges timed on the CPU oracle, ~15 s of CPU work (0 = skip)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="KEY=VALUE",
                    help="impg_gpu_set_option before the run (timing comparisons, e.g. locality_min=0)")
    ap.add_argument("--paf", default=None, help="reuse an existing synthetic PAF file")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path even with one rank")
    ap.add_argument("--workload", default="headline", choices=["headline", "config4", "config5", "skewed"],
                    help="headline: BASELINE.json's metric configuration (the default, what the driver runs).  config4: "
                         "HPRC-scale index from impg_synth_paf records (--records, default 5e7; 20 000 sequences), 100k ranges "
                         "-x -m 3.  config5: the headline index, contiguous 5 kb windows end to end (--ranges windows per GPU, "
                         "default the whole genome: 200 000), -x -m 5.  skewed: the NON-uniform index (impg_synth_skewed_paf_text: log-normal "
                         "alignment lengths 1 kb ... 5 Mb = 20 ... 10^5 ops a CIGAR, 1 %% of the sequences holding ~30 %% of the entries; "
                         "--records, default 2e5), --ranges (default 2 000) uniform 5 kb ranges, -x -m 3")
    ap.add_argument("--lanes", type=int, default=2, help="sharded index: chunks of the batch in flight at once per rank")
    ap.add_argument("--min-identity", type=float, default=None,
                    help="--min-result-identity of the reference (impg.rs:1283-1287); not part of the headline configuration")
    ap.add_argument("--no-extras", action="store_true", help="skip the full-results measurement (profiling runs)")
    ap.add_argument("--no-tiers", action="store_true", help="skip the checksum of the rows left in HBM and the other tiers' timings (profiling runs)")
    ap.add_argument("--form", default="rows", choices=["rows", "count", "ordered"],
                    help="what a timed step leaves behind.  rows (default, one GPU): every result row in HBM, attributable -- "
                         "impg_gpu_query_batch_device, 24 bytes a row (query id, four coordinates, the frontier record that names "
                         "its range and target).  count: nothing but the number of projections (impg_gpu_query_batch_stats without "
                         "per-range output: the final level's rows are written but not attributed) -- rounds 1-5's timed form, and "
                         "still the form of a sharded index.  ordered: the rows grouped by range in the reference's emission order "
                         "(IMPG_ROWS_ORDERED_SLOTS) -- the value_ordered_rows_device tier as the timed step, for profiling it")
    ap.add_argument("--world-sweep", action="store_true",
                    help="one device, constant work: the batch through a multi handle of 1 / 2 / 4 / 8 ranks that all share device 0 "
                         "(threads of this process, LocalComm) -- what the sharded path's fixed costs do as the world grows, "
                         "measurable without a second GPU; every world is checked against the plain index first")
    args = ap.parse_args()
    args.engine_option_changes_results = False  # (no engine option changes a result row: they are layout / schedule knobs)

    # `--gpus N` without a launcher: bring the N ranks up ourselves (one process per GPU) and pass the
    # one JSON line through.  Under torch.distributed.run the world must agree with --gpus.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run "
                 "--nproc-per-node %d bench.py --gpus %d ...) or drop WORLD_SIZE" %
                 (args.gpus, os.environ.get("WORLD_SIZE"), args.gpus, args.gpus))

    # stdout carries exactly one line, the result: anything a library prints to fd 1 (RCCL's version
    # banner does) is sent to stderr instead
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch

    import impg_amd
    wl = args.workload
    if args.records is None:
        args.records = 50_000_000 if wl == "config4" else 200_000 if wl == "skewed" else 1_000_000
    if args.ranges is None:
        args.ranges = 200_000 if wl == "config5" else 2_000 if wl == "skewed" else 100_000
    if args.max_depth is None:
        args.max_depth = 5 if wl == "config5" else 3
    sharded_run = args.gpus > 1 or args.force_sharded
    if args.chunk_ranges is None:
        # one GPU: the whole batch in one chunk (2.1e9 slots a level; every level's fixed costs paid once: 61.7 -> 59.5 ms per
        # step against two chunks of 50 000); a sharded index: two chunks per rank, one per lane, so that a lane's exchange
        # overlaps the other's kernels
        args.chunk_ranges = 500 if wl == "config5" else 2000 if wl == "skewed" else (50000 if sharded_run else max(50000, args.ranges))
    if args.pair_budget is None:
        # (config 5: 500 windows at depth 5 list 1.4e9 pairs in their last level -- under 2^30 every chunk was split after the run that
        # found it out, 797 ms per 4 000 windows against 608 with room for the level)
        args.pair_budget = (1 << 30) if sharded_run else (3 << 30)
    if wl != "headline":
        args.cpu_sample, args.no_extras = 0, True  # the CPU and full-results legs belong to the headline line

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
    n_seq, seq_len = (20000 if wl == "config4" else 200), 5_000_000
    paf = args.paf or os.path.join(tempfile.gettempdir(), "impg_synth%s_%d_seed42.paf" % ("_skewed" if wl == "skewed" else "", args.records))
    if wl == "config4":
        paf = None  # built straight from generated records: 31 GB of PAF text per 5e7 records would only be parsed again
    if paf and rank == 0 and not os.path.exists(paf):
        log("writing synthetic PAF %s" % paf)
        if wl == "skewed":
            impg_amd.synth_skewed_paf_text(paf + ".tmp", 42, args.records, n_seq=n_seq, seq_len=seq_len)
        else:
            impg_amd.synth_paf_text(paf + ".tmp", 42, args.records, n_seq=n_seq, seq_len=seq_len)
        os.replace(paf + ".tmp", paf)
    if dist is not None:
        dist.barrier()
        flush_c_stdio()  # (the communicator exists now: its banner, if any, goes out first)
    transitive = not args.no_transitive
    params = impg_amd.make_params(transitive=transitive, max_depth=args.max_depth)  # -x -m 3, defaults otherwise
    if args.min_identity is not None:
        params.min_identity = args.min_identity

    log("building the device index")
    t_build = time.time()
    comm = None
    comm_report = None
    if dist is not None:
        # this rank's shard of the index (targets bin-packed over the ranks); torch.distributed only carries the
        # RCCL ids to the ranks -- every collective of a query runs inside libimpg_gpu.so
        comm = impg_amd.Comm.rccl(rank, world, local_rank, lanes=args.lanes)
        comm.check()  # impg_gpu_comm_check on every lane (collective): known words through the all-gather before anything is built on it
        comm_report = {"kind": comm.kind(), "world": comm.world, "lanes": comm.lanes, "check": "ok on every lane"}
    t_gen = 0.0
    if paf:
        index = impg_amd.GpuImpg.from_paf(paf, device=local_rank, comm=comm)
    else:
        rec, ops, sl = impg_amd.synth_paf(42, args.records, n_seq=n_seq, seq_len=seq_len)
        t_gen = time.time() - t_build  # generating the records is not building the index
        log("synthetic records generated (%.1f s)" % t_gen)
        index = impg_amd.GpuImpg.from_records(rec, ops, sl, device=local_rank, comm=comm)
        del rec, ops
    t_build = time.time() - t_build - t_gen
    index.set_option("chunk_ranges", args.chunk_ranges)
    index.set_option("pair_budget", args.pair_budget)
    for kv in args.engine_option:
        k, v = kv.split("=", 1)
        index.set_option(k, int(v))

    # each rank is home to its own `--ranges` queries (weak scaling): seed 7 + rank
    ranges = np.zeros(args.ranges, dtype=impg_amd.RANGE_DTYPE)
    if wl == "config5":  # windows end to end over whole sequences; rank r tiles its own stretch of the genome
        per_seq = seq_len // 5000
        k = (np.arange(args.ranges, dtype=np.int64) + rank * args.ranges) % (per_seq * n_seq)
        ids = np.array([index.seq_id(impg_amd.synth_seq_name(t)) for t in range(n_seq)], dtype=np.uint32)
        ranges["target_id"], ranges["start"] = ids[k // per_seq], (k % per_seq) * 5000
        ranges["end"] = ranges["start"] + 5000
    else:
        bed = impg_amd.synth_bed(7 + rank, args.ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
        if paf:
            ids = np.array([index.seq_id(impg_amd.synth_seq_name(t)) for t in range(n_seq)], dtype=np.uint32)
            ranges["target_id"] = ids[bed["target_id"]]
        else:
            ranges["target_id"] = bed["target_id"]  # records carry the generator's ids
        ranges["start"], ranges["end"] = bed["start"], bed["end"]
    d_ranges = torch.from_numpy(ranges.view(np.uint8)).to(dev)  # resident in HBM before the timed region

    def step_count():  # (a collective call when the index is sharded: every rank brings its own ranges)
        st, _, _ = index.query_batch_stats(None, params, counts=False, checksums=False,
                                           device_ptr=d_ranges.data_ptr(), n=args.ranges)
        return st

    def step_rows():  # every row left in HBM, attributable; the handle's HBM goes back to the engine's pool for the next step
        dr = index.query_batch_device(None, params, device_ptr=d_ranges.data_ptr(), n=args.ranges)
        st = dr.stats
        dr.free()
        return st

    # (rows left on the device belong to one GPU's index; config 5's 10^12 rows do not fit any HBM: a caller would take them chunk by chunk)
    form = "count" if (dist is not None or wl in ("config5", "skewed")) else args.form
    def step_ordered():  # the rows grouped by range in the reference's emission order, in HBM (a profiling form: the tier's own step)
        import impg_amd._lib as _l
        dr = index.query_batch_device(None, params, device_ptr=d_ranges.data_ptr(), n=args.ranges, layout=_l.ROWS_ORDERED_SLOTS)
        st = dr.stats
        dr.free()
        return st

    step = step_rows if form == "rows" else step_ordered if form == "ordered" else step_count

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    log("index ready (%.1f s, %.2f GB in HBM); warmup" % (t_build, index.device_bytes() / 1e9))
    parity = None
    if dist is not None and paf:
        # Before anything is timed: the N-rank answer against the one-GPU answer.  Rank 0 submits its first 4096 ranges
        # (the other ranks none: the call is collective) with per-range counts and checksums, builds the PLAIN index of
        # the same PAF next to its shard, asks it the same, and the run goes on only if every count and checksum agrees.
        npar = min(4096, args.ranges)
        st_p, cnt_p, ck_p = index.query_batch_stats(ranges[:npar] if rank == 0 else ranges[:0], params)
        ok = 1
        if rank == 0:
            single = impg_amd.GpuImpg.from_paf(paf, device=local_rank)
            st_s, cnt_s, ck_s = single.query_batch_stats(ranges[:npar], params)
            del single
            ok = int(st_p.projected == st_s.projected and np.array_equal(cnt_p, cnt_s) and np.array_equal(ck_p, ck_s))
            parity = {"parity_vs_single": bool(ok), "ranges_checked": npar, "projected_sharded": int(st_p.projected),
                      "projected_single": int(st_s.projected)}
            log("parity vs the single-GPU index on %d ranges: %s (%d projections)" % (npar, "identical" if ok else "DIFFERENT", st_s.projected))
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.broadcast(flag, src=0)
        if int(flag.item()) != 1:
            if rank == 0:
                result_out.write(json.dumps({"metric": "projected ranges/sec, 1M-PAF 100k-BED -x depth 3; CPU coitrees baseline", "value": None,
                                             "n_gpus": world, "error": "the %d-rank answer differs from the single-GPU answer" % world,
                                             **parity}) + "\n")
                result_out.flush()
            sys.exit(3)
    if world > 1 or args.force_sharded:
        # part of the set-up, like the index build: the first collective pass on a fresh process brings up the
        # transport (RCCL channels, peer mappings) and sizes every lane's scratch -- seconds on a cold box, and with
        # two chunks in flight per rank one --warmup step does not always touch both lanes' buffers
        # (passes until two in a row take about the same time, at most six: on a cold box the first ones also pay for the
        # device's page mappings of every lane's buffers, 1.1 s then 0.45 s where a settled pass takes 0.06 s)
        prev = None
        for k in range(6):
            ts = time.perf_counter()
            st = step()
            cur = time.perf_counter() - ts
            log("transport + scratch priming pass %d: %.1f ms wall, %.1f ms of engine time" % (k + 1, cur * 1e3, st.ms_total))
            if prev is not None and cur > 0.9 * prev:
                break
            prev = cur
    for _ in range(args.warmup):
        st = step()
        log("warmup step: %d projected, engine %.1f ms (lookup %.1f project %.1f update %.1f exchange %.1f)" %
            (st.projected, st.ms_total, st.ms_lookup, st.ms_project, st.ms_update, st.ms_exchange))
    if dist is not None:
        index.hop_profile(reset=True)
    sync()
    t0 = time.perf_counter()
    stats, step_ms = [], []
    for _ in range(args.steps):  # (a step returns when its results are complete: the per-step clock adds no synchronisation)
        ts = time.perf_counter()
        stats.append(step())
        step_ms.append((time.perf_counter() - ts) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    log("timed region: %.3f s for %d steps (%s ms)" % (dt, args.steps, " ".join("%.1f" % x for x in step_ms)))
    hops = sharded_diagnostics(index, stats, args, dist, dev, world) if dist is not None else None
    strong = strong_scaling_leg(index, args, params, dist, dev, rank, world, local_rank, n_seq, seq_len, bool(paf), sync) \
        if (dist is not None and wl == "headline" and not args.no_extras) else None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(sum(s.projected for s in stats))], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        projected_total = float(tot.item())
    else:
        projected_total = float(sum(s.projected for s in stats))

    run_multi = dist is not None and wl == "headline" and not args.no_extras and paf
    wait_group = dist.new_group(backend="gloo") if run_multi else None  # (a CPU-side wait: an RCCL barrier would spin on the other GPUs)
    if rank != 0:
        if dist is not None:
            del index
            comm.close()
            if run_multi:
                dist.barrier(group=wait_group)  # the shards are gone: rank 0 may build the multi handle over every GPU
                dist.barrier(group=wait_group)  # ... and is done with it
            dist.destroy_process_group()
        flush_c_stdio()
        return

    proj_per_step = sum(s.projected for s in stats) / max(1, args.steps)
    self_check = self_check_timed(stats, args, wl, world, dist is not None)
    if wl != "headline" and dist is None and self_check["status"] == "ok":
        # no pinned constant for these workloads: the timed form (no per-range output, fused final level) and the counting
        # form (per-range counts and checksums, the form the suite compares with the oracle) on the same sample of the
        # batch must project the same ranges, and the counts must add up to it
        ns = min(args.ranges, 2000)
        st_t, _, _ = index.query_batch_stats(ranges[:ns], params, counts=False, checksums=False)
        st_c, cnt_c, _ = index.query_batch_stats(ranges[:ns], params)
        self_check["sample"] = {"ranges": ns, "timed_form_projected": int(st_t.projected), "counting_form_projected": int(st_c.projected),
                                "sum_of_per_range_counts": int(cnt_c.sum())}
        if not (st_t.projected == st_c.projected == int(cnt_c.sum())):
            self_check["status"] = "FAILED: the timed form and the counting form disagree on the first %d ranges" % ns
    tiers = None
    if dist is None and form == "rows" and not args.no_tiers:
        tiers = rows_form_legs(index, ranges, params, args, d_ranges, step_count, sync, self_check, wl, dt)
    ms_project = sum(s.ms_project for s in stats)
    launches = sum(s.project_launches for s in stats)
    ach = (sum(s.projected for s in stats) * ALG_BYTES_PER_PROJECTION) / (ms_project * 1e-3) / 1e9 if ms_project > 0 else 0.0
    # HBM bytes per projection measured with rocprofv3 PMC passes of this same command
    # (scripts/profile_r1.sh -> profiles/r1_v2_traffic.json; FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
    traffic = None
    # the counters of THIS workload's profile (headline: r*_final_*; config 4 / 5: r*_config4_* / r*_config5_*), and only when
    # the run is the profiled configuration -- another index size or the identity filter moves other bytes per pair
    tag = "final" if wl == "headline" else wl
    if wl == "config4" and args.records == 100_000_000:
        tag = "config4_1e8"  # (BASELINE config 4's full 10^8 records: its own profile, profiles/r6_config4_1e8_*)
    profiled = args.min_identity is None and args.records == (50_000_000 if tag == "config4" else 100_000_000 if tag == "config4_1e8" else 1_000_000)
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_traffic.json" % tag))) if profiled else []  # newest round last
    tpath = tfiles[-1] if tfiles else ""
    if os.path.exists(tpath) and launches:
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj["hbm_bytes_per_pair"] * (sum(s.pairs for s in stats) or sum(s.projected for s in stats)) / launches
    out = {
        "metric": "projected ranges/sec, 1M-PAF 100k-BED -x depth 3; CPU coitrees baseline" if wl == "headline" else
                  "projected ranges/sec, BASELINE %s (not the headline metric)" % wl,
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
            "workload": "synthetic PAF %d records (%d seqs x 5 Mb, %s, bidirectional "
                        "index), %d %s per GPU, %s" %
                        (args.records, n_seq, "log-normal alignment lengths 1 kb ... 5 Mb (median 8 kb; one {= run, edit} block per 100 bases: 20 ... 10^5 ops "
                                              "a CIGAR), 1 % of the sequences target / query of 30 % of the records" if wl == "skewed" else
                                              "10 kb alignments, 200-op CIGARs", args.ranges, "contiguous 5 kb windows" if wl == "config5" else "x 5 kb query ranges",
                         ("-x -m %d" % args.max_depth) if transitive else "no transitive"),
            "records": args.records, "ranges_per_gpu": args.ranges, "max_depth": args.max_depth if transitive else 0,
            "min_transitive_len": 101, "min_distance_between_ranges": 10, "min_identity": args.min_identity,
            "parallelism": ("1 process/GPU, index sharded by target sequence (bin-packed), frontier all-to-all-v over RCCL, "
                            "%d chunks in flight per rank" % args.lanes) if dist is not None else "single GPU",
            "chunk_ranges": args.chunk_ranges, "pair_budget": args.pair_budget,
        },
        "projected_per_step_rank0": proj_per_step,
        "pairs_per_step_rank0": sum(s.pairs for s in stats) / max(1, args.steps),
        "frontier_ranges_per_step_rank0": sum(s.frontier_ranges for s in stats) / max(1, args.steps),
        "stage_ms_per_step_rank0": {"lookup": sum(s.ms_lookup for s in stats) / max(1, args.steps),
                                    "project": ms_project / max(1, args.steps),
                                    "update": sum(s.ms_update for s in stats) / max(1, args.steps),
                                    "exchange_wall": sum(s.ms_exchange for s in stats) / max(1, args.steps),
                                    "engine_total": sum(s.ms_total for s in stats) / max(1, args.steps)},
        "step_ms_rank0": step_ms,
        "argv": sys.argv[1:],
        "index_build_s": t_build,
        "index_bytes": index.device_bytes(),
        "roofline": roofline(stats, ach, traffic, tpath, ms_project, launches, tag if profiled else None),
        "self_check": self_check["status"],
        "self_check_detail": self_check,
        "timed_form": ("rows: every result row left in HBM, 24 B each -- query id, q_first, q_last, t_first, t_last and the frontier record that "
                       "names its range of the batch and its target (impg_gpu_query_batch_device, IMPG_ROWS_ATTRIBUTED); slot order free")
                      if form == "rows" else
                      "ordered: every slot's row at its place of the batch's ordered output, 24 B each, grouped by range in the reference's emission "
                      "order, hole rows where a projection returned None (impg_gpu_query_batch_device, IMPG_ROWS_ORDERED_SLOTS)"
                      if form == "ordered" else
                      "count: the number of projections only (impg_gpu_query_batch_stats without per-range output; the final level's rows "
                      "are computed and stored but not attributed to a range)",
    }
    if tiers:
        out.update(tiers)
    if dist is not None:
        out["comm"] = comm_report
        out["parity"] = parity
        out["parity_vs_single"] = parity["parity_vs_single"] if parity else None
        out["hops"] = hops
        out["strong_scaling"] = strong
    if world == 1 and dist is None and args.world_sweep:
        out["world_sweep"] = world_sweep_leg(index, paf, ranges, params, args)
    if world == 1 and dist is None and not args.no_extras:
        out["full_results"] = full_results_leg(index, ranges, params)
        out["dfs_batch"] = dfs_batch_leg(index, ranges, args.max_depth)
        # the rows the trait would return for the same batch: one per projection plus every range's self interval
        sp = out["full_results"]["stream"]
        self_check["stream_projected"] = sp["projected"]
        self_check["stream_rows_minus_self"] = sp["rows"] - len(ranges)
        if wl == "headline" and self_check["status"] == "ok" and not (
                sp["projected"] == self_check["timed_projected_per_step"] == sp["rows"] - len(ranges)):
            self_check["status"] = out["self_check"] = "FAILED: the row stream's projections differ from the timed form's"
    out["cpu_baseline"] = cpu_baseline(args, paf, ranges, transitive) if (world == 1 and args.cpu_sample > 0) else None
    if dist is not None:
        del index
        comm.close()
        if run_multi:
            dist.barrier(group=wait_group)
            try:
                out["multi_handle"] = multi_handle_leg(paf, ranges, params, args, world)
            except Exception as e:  # (a diagnostic leg: its failure is reported, the headline line still goes out)
                out["multi_handle"] = {"error": str(e)}
            dist.barrier(group=wait_group)
        dist.destroy_process_group()
    flush_c_stdio()
    result_out.write(json.dumps(out) + "\n")
    result_out.flush()
    if out["self_check"] != "ok":  # a number whose answer is wrong is not a measurement
        log("SELF CHECK FAILED: %s" % out["self_check"])
        sys.exit(4)


def self_check_timed(stats, args, wl, world, sharded):
    """What the timed steps computed, against what they must compute: every step the same total; on the headline
    configuration of one rank the pinned constant (the sharded runs have their own parity_vs_single leg)."""
    per = sorted({int(s.projected) for s in stats})
    chk = {"status": "ok", "timed_projected_per_step": per[0] if len(per) == 1 else per, "expected": None}
    if len(per) != 1:
        chk["status"] = "FAILED: the timed steps disagree with each other"
        return chk
    pinned = (wl == "headline" and not args.no_transitive and args.max_depth == 3 and args.records == 1_000_000 and
              args.ranges == 100_000 and args.min_identity is None and not args.paf and not args.engine_option_changes_results)
    if pinned and world == 1:  # (one rank through the sharded path answers the same batch; N ranks bring N different batches)
        chk["expected"] = HEADLINE_PROJECTED
        if per[0] != HEADLINE_PROJECTED:
            chk["status"] = "FAILED: %d projections per step, the headline batch has %d" % (per[0], HEADLINE_PROJECTED)
    return chk


def rows_form_legs(index, ranges, params, args, d_ranges, step_count, sync, self_check, wl, dt_rows):
    """The timed form leaves rows; this says (a) that they are the right rows -- per-range counts and order-independent
    checksums recomputed FROM THE ROWS IN HBM (impg_gpu_device_rows_check: every slot attributed through its frontier
    record), their sums against the pinned constants at the headline, against the counting form's otherwise -- and (b)
    what the other tiers of the same batch cost: nothing attributed (rounds 1-5's timed form) and, at the headline,
    rows grouped by range in the reference's emission order."""
    import numpy as np
    out = {}
    dr = index.query_batch_device(None, params, device_ptr=d_ranges.data_ptr(), n=args.ranges)
    cnt, ck = dr.check()
    parts = [{"level": int(d.level), "slots": int(d.n_slots), "frontier": int(d.n_frontier)} for d in dr.parts()]
    projected = dr.projected
    dr.free()
    with np.errstate(over="ignore"):
        rows_sum = int(ck.sum(dtype=np.uint64))
    rows_cnt = int(cnt.sum())
    self_check["rows_in_hbm"] = {"sum_of_per_range_counts": rows_cnt, "sum_of_per_range_checksums": rows_sum, "projected": int(projected),
                                 "parts": parts, "bytes_per_slot": 24}
    if self_check["status"] == "ok" and rows_cnt != projected:
        self_check["status"] = "FAILED: the rows left in HBM (%d) are not the projections counted (%d)" % (rows_cnt, projected)
    if self_check.get("expected") is not None:  # the headline batch: pinned
        self_check["expected_checksum"] = HEADLINE_CHECKSUM
        if self_check["status"] == "ok" and HEADLINE_CHECKSUM is not None and rows_sum != HEADLINE_CHECKSUM:
            self_check["status"] = "FAILED: checksum of the rows left in HBM %d, the headline batch's is %d" % (rows_sum, HEADLINE_CHECKSUM)
    elif self_check["status"] == "ok":
        ns = min(args.ranges, 2000)
        st_c, cnt_c, ck_c = index.query_batch_stats(ranges[:ns], params)
        ok = bool((cnt_c == cnt[:ns]).all() and (ck_c == ck[:ns]).all())
        self_check["rows_vs_counting_form"] = {"ranges": ns, "identical": ok}
        if not ok:
            self_check["status"] = "FAILED: rows left in HBM and the counting form disagree on the first %d ranges" % ns
    # ---- the other tiers of the same batch --------------------------------------------------------------------------
    for _ in range(max(1, args.warmup)):
        step_count()
    sync()
    t0 = time.perf_counter()
    sts = [step_count() for _ in range(args.steps)]
    sync()
    dtc = time.perf_counter() - t0
    pc = sum(s.projected for s in sts)
    out["value_count_only"] = pc / dtc if dtc > 0 else None
    out["ms_per_step_count_only"] = dtc * 1e3 / max(1, args.steps)
    out["count_only_stage_ms"] = {"lookup": sum(s.ms_lookup for s in sts) / max(1, args.steps), "project": sum(s.ms_project for s in sts) / max(1, args.steps),
                                  "update": sum(s.ms_update for s in sts) / max(1, args.steps)}
    if wl == "headline":
        # rows grouped by range in the reference's emission order, left in HBM (IMPG_ROWS_ORDERED): engine + row placement, no D2H
        import impg_amd
        def step_ordered(layout):
            do = index.query_batch_device(None, params, device_ptr=d_ranges.data_ptr(), n=args.ranges, layout=layout)
            st, pm, rows = do.stats, do.place_ms, sum(int(d.n_slots) for d in do.parts())
            do.free()
            return st, pm, rows

        def ordered_leg(layout, steps, what):
            for _ in range(max(1, args.warmup)):
                step_ordered(layout)
            sync()
            t0 = time.perf_counter()
            so = [step_ordered(layout) for _ in range(steps)]
            sync()
            dto = time.perf_counter() - t0
            po = sum(s[0].projected for s in so)
            return po, steps, {"value": po / dto if dto > 0 else None, "ms_per_step": dto * 1e3 / max(1, steps),
                               "rows_per_step": so[-1][2], "engine_ms": sum(s[0].ms_total for s in so) / max(1, steps),
                               "placement_ms": sum(s[1] for s in so) / max(1, steps),
                               "stage_ms": {"lookup": sum(s[0].ms_lookup for s in so) / max(1, steps),
                                            "project": sum(s[0].ms_project for s in so) / max(1, steps),
                                            "update": sum(s[0].ms_update for s in so) / max(1, steps)}, "what": what}
        po, ks, leg = ordered_leg(impg_amd._lib.ROWS_ORDERED_SLOTS, args.steps,
                                  "impg_gpu_query_batch_device(IMPG_ROWS_ORDERED_SLOTS): the batch's rows grouped by range in the reference's emission "
                                  "order (self interval, then level by level in frontier order x visit order), 24 B each, every slot at its place -- a "
                                  "None projection is a hole row (query_id 0xFFFFFFFF) -- and per-range offsets, in HBM; the final level's kernel "
                                  "writes its rows itself where they belong; wall time per step, no D2H")
        out["value_ordered_rows_device"] = leg["value"]
        out["ms_per_step_ordered_rows_device"] = leg["ms_per_step"]
        out["ordered_rows_device"] = leg
        pc_per = pc / max(1, args.steps)
        if po / max(1, ks) != pc_per:
            self_check["status"] = "FAILED: the ordered rows' projections differ from the count-only form's"
        log("tiers: ordered rows in HBM (slots) %.2f ms/step (engine %.2f, of which placement %.2f)" % (leg["ms_per_step"], leg["engine_ms"], leg["placement_ms"]))
        if not args.no_extras:
            po2, ks2, leg2 = ordered_leg(impg_amd._lib.ROWS_ORDERED, 1,
                                         "impg_gpu_query_batch_device(IMPG_ROWS_ORDERED): the same rows with the holes closed (what impg_gpu_query_batch "
                                         "returns, left in HBM): every level kept and listed, rows placed by rows_device.hip's scans and scatters")
            out["ordered_rows_device_compact"] = leg2
            if po2 / max(1, ks2) != pc_per:
                self_check["status"] = "FAILED: the compact ordered rows' projections differ from the count-only form's"
    out["tiers_note"] = ("value = rows left in HBM, attributable (the timed form); value_count_only = the same batch with nothing but the "
                         "projection count kept (rounds 1-5's timed form); value_ordered_rows_device = the same rows grouped by range in the "
                         "reference's emission order, left in HBM; full_results.stream = those delivered to the host (PCIe-bound)")
    log("tiers: rows in HBM %.2f ms/step, count only %.2f ms/step" % (dt_rows * 1e3 / max(1, args.steps), out["ms_per_step_count_only"]))
    return out


SIMDS = 256 * 4


def roofline(stats, ach, traffic, tpath, ms_project, launches, tag):
    """The contractual HBM accounting (856 algorithmic bytes per projection) next to what the memory
    system and the issue ports actually did for project_kernel: measured HBM traffic as a fraction of peak
    (profiles/r*_traffic.json) and the VALU-issue fraction (profiles/r*_sq.json: SQ_INSTS_VALU x the measured cost of
    the kernel's instruction mix, scripts/issue_rate.hip + scripts/valu_mix.py -- SQ_ACTIVE_INST_VALU ticks once per
    instruction whatever it costs, so a flat 2 or 4 clocks cannot be read off the counters)."""
    avg_ms = ms_project / launches if launches else None
    r = {
        "kernel": "the projection launches of a step: project_entries_kernel (the dense final level, ~94 % of the pairs), "
                  "project_staged_kernel (dense listed levels), project_kernel (sparse levels)",
        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS,
        "frac_note": "the contract's accounting: 856 algorithmic bytes per projection (a streamed 200-op CIGAR) / launch time / 8 TB/s. "
                     "The kernels read a 64-byte entry and <= 2 x 64 bytes of prefix line per pair instead -- on a dense level from "
                     "copies staged in LDS once per ~240 pairs -- so this exceeds 1 and is not a bandwidth; the physical fractions "
                     "are measured_traffic_frac and valu_issue_frac",
        "traffic": traffic,
        "traffic_note": "bytes per launch = rocprofv3 (FETCH_SIZE x2 [gfx950] + WRITE_SIZE) per pair, from "
                        "profiles/%s, x pairs per launch of this run" % os.path.basename(tpath),
        "algorithmic_bytes_per_projection": ALG_BYTES_PER_PROJECTION,
        "launches": launches,
        "avg_launch_ms": avg_ms,
        "avg_projections_per_launch": (sum(s.projected for s in stats) / launches) if launches else None,
    }
    tgbs = (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms) else None
    r["achieved_traffic_GBs"] = tgbs
    r["measured_traffic_frac"] = (tgbs / HBM_PEAK_GBS) if tgbs else None
    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_sq.json" % tag))) if tag else []
    vf = None
    if sq:
        with open(sq[-1]) as f:
            j = json.load(f)
        vf = j.get("valu_issue_frac")
        r["valu_issue_frac"] = vf
        r["valu_insts_per_pair"] = j.get("valu_insts_per_pair")
        r["valu_cycles_per_inst"] = j.get("cycles_per_valu_inst")
        r["valu_note"] = ("SQ_INSTS_VALU x %.2f clocks (the measured issue cost of the kernel's instruction mix: profiles/r3_issue_rate.json, "
                          "profiles/r*_valu_mix_*.json) / (GRBM_GUI_ACTIVE x %d SIMDs) of the projection kernels, profiles/%s" %
                          (j.get("cycles_per_valu_inst") or 0.0, SIMDS, os.path.basename(sq[-1])))
    # every kernel of the step that takes >= 1 % of it (scripts/make_per_kernel_json.py on the PMC passes of this command)
    pk = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_per_kernel.json" % tag))) if tag else []
    if pk:
        with open(pk[-1]) as f:
            j = json.load(f)
        r["per_kernel"] = [{"name": k["name"], "ms": k["ms_per_step"], "hbm_frac": k.get("hbm_frac"), "valu_issue_frac": k.get("valu_issue_frac"),
                            "waves_per_simd": k.get("waves_per_simd_mean")} for k in j["kernels"]]
        r["per_kernel_source"] = "profiles/" + os.path.basename(pk[-1])
    mf = r["measured_traffic_frac"]
    # what binds, from the two measured fractions: neither HBM (the 856-byte model's bound) nor the VALUs are saturated on the
    # headline index -- the rest is latency the resident waves do not cover (dependent 16-byte gathers, each lane its own line)
    if mf is not None and vf is not None:
        r["bound"] = "hbm" if mf >= 0.6 else ("valu-issue" if vf >= 0.75 else "latency (HBM at %.2f, VALU issue at %.2f of their peaks)" % (mf, vf))
    else:
        r["bound"] = "unmeasured (no PMC summaries of this configuration under profiles/)"
    r["contract_bound"] = "hbm"
    r["limiter"] = ("on the headline index the projection is bound by VALU issue: the final level runs entry by entry with the index "
                    "staged in LDS (project_entries_kernel, DESIGN.md 5.2 item 19), HBM sees little more than the result stores "
                    "(measured_traffic_frac), the vector ALUs issue valu_issue_frac of their slots at 4 waves per SIMD.  On the config-4 "
                    "index (20 000 sequences, no reuse: project_kernel) the projection IS HBM-bound: 292 B per pair = 4.9 TB/s of "
                    "128-byte gathers, 0.62 of peak (profiles/r3_config4_traffic.json); DESIGN.md 5.2, 7")
    return r


def sharded_diagnostics(index, stats, args, dist, dev, world):
    """Per-hop breakdown of the timed steps on a sharded index: every rank's impg_gpu_index_hop_profile (wall seconds of
    the hop's stages and bytes sent, by hop number, summed over its lanes) and its engine's stage times, gathered to
    rank 0 and reported per step as max and mean over the ranks -- what says whether an N-GPU step waits for compute, for
    the links, or for its slowest rank."""
    import numpy as np
    import torch
    import impg_amd
    prof = index.hop_profile(reset=True)[0]  # [8][12]
    eng = np.array([sum(s.ms_lookup for s in stats), sum(s.ms_project for s in stats), sum(s.ms_update for s in stats),
                    sum(s.ms_exchange for s in stats), sum(s.ms_total for s in stats), float(sum(s.projected for s in stats)),
                    float(sum(s.pairs for s in stats))], dtype=np.float64)
    mine = torch.from_numpy(np.concatenate([prof.reshape(-1), eng])).to(dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    a = np.stack([t.cpu().numpy() for t in allr])  # [world][96 + 7]
    P = a[:, :96].reshape(world, 8, 12) / max(1, args.steps)
    E = a[:, 96:] / max(1, args.steps)
    F = impg_amd.index.HOP_PROFILE_FIELDS
    hops = []
    for h in range(8):
        if P[:, h, 0].max() <= 0:
            continue
        hops.append({"hop": h + 1, **{F[f]: {"max": float(P[:, h, f].max()), "mean": float(P[:, h, f].mean())} for f in range(12)}})
    names = ["ms_lookup", "ms_project", "ms_update", "ms_exchange_wall", "ms_engine_total", "projected", "pairs"]
    return {"per_step": True, "note": "seconds are host wall time per step summed over a rank's lanes (a stage that ends in a collective includes "
                                      "the wait for the slowest rank); max / mean over the ranks",
            "by_hop": hops, "engine_per_rank": {names[k]: {"max": float(E[:, k].max()), "mean": float(E[:, k].mean()), "min": float(E[:, k].min())}
                                                for k in range(len(names))}}


def strong_scaling_leg(index, args, params, dist, dev, rank, world, local_rank, n_seq, seq_len, from_paf, sync):
    """The SAME batch whatever the world: rank 0's 100 000 ranges (seed 7) dealt out to the ranks, every n-th to rank n --
    total work fixed, next to the weak-scaling headline where every rank brings its own 100 000."""
    import numpy as np
    import torch
    import impg_amd
    bed = impg_amd.synth_bed(7, args.ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)[rank::world]
    r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
    if from_paf:
        ids = np.array([index.seq_id(impg_amd.synth_seq_name(t)) for t in range(n_seq)], dtype=np.uint32)
        r["target_id"] = ids[bed["target_id"]]
    else:
        r["target_id"] = bed["target_id"]
    r["start"], r["end"] = bed["start"], bed["end"]
    d_r = torch.from_numpy(r.view(np.uint8)).to(dev)

    def step():
        st, _, _ = index.query_batch_stats(None, params, counts=False, checksums=False, device_ptr=d_r.data_ptr(), n=len(r))
        return st
    step()
    sync()
    t0 = time.perf_counter()
    sts = [step() for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(sum(s.projected for s in sts))], dtype=torch.float64, device=dev)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dtm, proj = float(tmax[0].item()), float(t[1].item())
    return {"workload": "the headline's %d ranges (seed 7) dealt out over the %d ranks" % (args.ranges, world), "scaling": "strong",
            "value": proj / dtm if dtm > 0 else None, "unit": "projected ranges/s", "ms_per_step": dtm * 1e3 / max(1, args.steps),
            "projected_per_step": proj / max(1, args.steps)}


def multi_handle_leg(paf, ranges, params, args, world):
    """One process, one handle over all the node's GPUs (impg_gpu_index_create_from_paf_multi: LocalComm, peer copies over
    xGMI, no RCCL): rank 0's batch through it, after the ranks have let go of their shards."""
    import impg_amd
    log("multi-handle leg: one process over %d GPU(s)" % world)
    t0 = time.perf_counter()
    mh = impg_amd.GpuImpg.from_paf(paf, devices=list(range(world)), lanes=args.lanes)
    build_s = time.perf_counter() - t0
    mh.set_option("chunk_ranges", max(1, min(50000, (len(ranges) + world - 1) // world)))
    mh.set_option("pair_budget", 1 << 30)
    mh.query_batch_stats(ranges, params, counts=False, checksums=False)
    mh.query_batch_stats(ranges, params, counts=False, checksums=False)
    mh.hop_profile(reset=True)
    t0 = time.perf_counter()
    sts = [mh.query_batch_stats(ranges, params, counts=False, checksums=False)[0] for _ in range(args.steps)]
    dt = time.perf_counter() - t0
    proj = sum(s.projected for s in sts)
    prof = mh.hop_profile(reset=True) / max(1, args.steps)  # [world][8][12]
    F = impg_amd.index.HOP_PROFILE_FIELDS
    hops = [{"hop": h + 1, **{F[f]: {"max": float(prof[:, h, f].max()), "mean": float(prof[:, h, f].mean())} for f in range(12)}}
            for h in range(8) if prof[:, h, 0].max() > 0]
    del mh
    return {"workload": "rank 0's %d ranges through ONE handle over %d GPU(s) of this process (ranges from host memory; strong scaling)" % (len(ranges), world),
            "value": proj / dt if dt > 0 else None, "unit": "projected ranges/s", "ms_per_step": dt * 1e3 / max(1, args.steps),
            "projected_per_step": proj / max(1, args.steps), "index_build_s": build_s, "by_hop": hops}


def world_sweep_leg(plain, paf, ranges, params, args):
    """The SAME batch through the sharded path with 1, 2, 4 and 8 ranks that share this one device (a multi handle over
    devices [0] * w: one thread per rank, LocalComm, peer copies that stay on the device): total work is constant, the
    kernels of all ranks run on the same GPU, so whatever ms_per_step gains with w is the sharded path's own cost per rank
    and hop -- routing, the two all-gathers, pack / unpack, the host synchronisations -- which an N-GPU run pays too,
    there in parallel.  Every world's per-range counts and checksums are compared with the plain index's first."""
    import numpy as np
    import impg_amd
    if not paf:
        return {"error": "needs the PAF (headline / config 5 workloads)"}
    npar = min(4096, len(ranges))
    plain.set_option("chunk_ranges", 4096)
    st_s, cnt_s, ck_s = plain.query_batch_stats(ranges[:npar], params)
    plain.set_option("chunk_ranges", args.chunk_ranges)
    F = impg_amd.index.HOP_PROFILE_FIELDS
    legs = []
    for w in (1, 2, 4, 8):
        log("world sweep: %d rank(s) on device 0" % w)
        t0 = time.perf_counter()
        mh = impg_amd.GpuImpg.from_paf(paf, devices=[0] * w, lanes=args.lanes)
        build_s = time.perf_counter() - t0
        mh.set_option("pair_budget", 1 << 30)
        mh.set_option("chunk_ranges", 4096)
        st_p, cnt_p, ck_p = mh.query_batch_stats(ranges[:npar], params)
        ok = bool(st_p.projected == st_s.projected and np.array_equal(cnt_p, cnt_s) and np.array_equal(ck_p, ck_s))
        # (the batch is dealt to the ranks in contiguous blocks; a rank's block in chunks of at most 50 000, one per lane at w = 1)
        mh.set_option("chunk_ranges", max(1, min(50000, (len(ranges) + w * args.lanes - 1) // (w * args.lanes))))
        for _ in range(2):
            mh.query_batch_stats(ranges, params, counts=False, checksums=False)
        mh.hop_profile(reset=True)
        t0 = time.perf_counter()
        sts = [mh.query_batch_stats(ranges, params, counts=False, checksums=False)[0] for _ in range(args.steps)]
        dt = time.perf_counter() - t0
        prof = mh.hop_profile(reset=True) / max(1, args.steps)  # [w][8][12]
        hops = int((prof[:, :, 0].sum(axis=0) > 0).sum())
        per_rank = prof.sum(axis=1)  # [w][12]: a rank's seconds per step over its hops and lanes
        fixed = {F[f]: float(per_rank[:, f].sum()) * 1e3 for f in (1, 2, 3, 5, 6, 7)}  # everything but the owner's expansion, ms summed over ranks
        legs.append({"world": w, "parity_vs_single": ok, "ms_per_step": dt * 1e3 / max(1, args.steps),
                     "projected_per_step": sum(s.projected for s in sts) / max(1, args.steps), "index_build_s": build_s, "hops": hops,
                     "owner_expand_ms_sum_over_ranks": float(per_rank[:, 4].sum()) * 1e3,
                     "other_stages_ms_sum_over_ranks": fixed,
                     "fixed_ms_per_rank_and_hop": float(sum(fixed.values())) / max(1, w * max(1, hops))})
        del mh
    base = legs[0]["ms_per_step"]
    return {"workload": "the batch's %d ranges through ONE multi handle of w ranks sharing device 0 (LocalComm, %d lanes a rank); constant work" %
                        (len(ranges), args.lanes),
            "legs": legs, "growth_1_to_8": legs[-1]["ms_per_step"] / base if base else None,
            "parity_vs_single_all": all(l["parity_vs_single"] for l in legs),
            "note": "seconds of the stages are host wall time summed over a rank's lanes (stages that end in an exchange include the wait "
                    "for the other ranks, which here share the device)"}


def full_results_leg(index, ranges, params):
    """What the trait returns: impg_gpu_query_batch with every result row placed on the device (rows_device.hip)
    and copied back once into a pinned block (BASELINE config 3: the first 10 000 ranges, same flags).  Not the
    headline value: 2.1e9 rows x 24 B per headline step cannot cross PCIe at the engine's rate.  Two calls: the
    first pins its 5 GB result block (hipHostMalloc), the second reuses it from the library's pool, which is the
    state of a process that serves more than one batch."""
    n = min(10_000, len(ranges))
    legs = []
    for label in ("first call (result block pinned during the call)", "second call (result block recycled)"):
        log("full-results leg: impg_gpu_query_batch on %d ranges, %s" % (n, label))
        t0 = time.perf_counter()
        res = index.query_batch(ranges[:n], params, copy=False)
        dt = time.perf_counter() - t0
        rows, proj = res.total, res.projected
        eng, asm = res.timing()
        del res
        legs.append({"call": label, "rows": rows, "projected": proj, "seconds": dt, "projected_per_s": proj / dt if dt > 0 else None,
                     "engine_s": eng, "assemble_s": asm})
    out = dict(legs[-1])
    out["workload"] = ("config 3: first %d ranges, same PAF and flags, impg_gpu_query_batch (rows placed on the device, one D2H "
                       "into a pinned result block)" % n)
    out["calls"] = legs
    out["stream"] = row_stream_leg(index, ranges, params)
    return out


def row_stream_leg(index, ranges, params):
    """The whole headline batch's rows through impg_gpu_query_batch_stream: 4 000-range chunks (2 GB of rows each) computed on two engines in
    turn, each chunk's rows copied into one of two pinned blocks while the next chunk is computed, handed over in range
    order.  The consumer here reads one column of every chunk (the rows are touched, not just counted).  Two passes: the
    first pins the two blocks."""
    import numpy as np
    chunk = 4_000
    legs = []
    for label in ("first call (the two blocks pinned during the call)", "second call (blocks recycled)"):
        log("row-stream leg: impg_gpu_query_batch_stream on %d ranges, %s" % (len(ranges), label))
        seen = {"rows": 0, "chunks": 0, "ranges": 0, "sum": 0, "max_chunk_rows": 0}

        def consume(first, part):
            assert first == seen["ranges"]
            seen["rows"] += part.total
            seen["ranges"] += len(part)
            seen["chunks"] += 1
            seen["max_chunk_rows"] = max(seen["max_chunk_rows"], part.total)
            if part.total:
                seen["sum"] += int(part.intervals["query_id"][:: max(1, part.total // 65536)].astype(np.int64).sum())
            return False

        t0 = time.perf_counter()
        proj = index.query_batch_stream(ranges, consume, params, chunk_ranges=chunk)
        dt = time.perf_counter() - t0
        legs.append({"call": label, "rows": seen["rows"], "projected": proj, "chunks": seen["chunks"], "seconds": dt,
                     "rows_per_s": seen["rows"] / dt if dt > 0 else None, "GB_per_s": seen["rows"] * 24 / dt / 1e9 if dt > 0 else None,
                     "largest_chunk_GB": seen["max_chunk_rows"] * 24 / 1e9})
    out = dict(legs[-1])
    out["workload"] = ("the headline batch's rows (%d ranges, -x -m 3) through impg_gpu_query_batch_stream, %d-range chunks, "
                       "two pinned blocks" % (len(ranges), chunk))
    out["calls"] = legs
    return out


def dfs_batch_leg(index, ranges, max_depth):
    """`--transitive-dfs` for the first 10 000 ranges, counting form: the per-query walk kernel (walk_device.inc), one
    launch for the batch.  (Round 2's batch engine paid a launch sequence per pop round: minutes for this batch.)"""
    import impg_amd
    n = min(10_000, len(ranges))
    p = impg_amd.make_params(transitive=True, dfs=True, max_depth=max_depth)
    # two calls: the first allocates the per-wave slabs of the walk (15 GB for a batch that fills the device's wave slots),
    # the second is a process that has served a DFS batch before
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        st, _, _ = index.query_batch_stats(ranges[:n], p, counts=False, checksums=False)
        times.append(time.perf_counter() - t0)
    dt = times[-1]
    return {"workload": "first %d ranges, --transitive-dfs -m %d, counting form" % (n, max_depth), "projected": st.projected, "seconds": dt,
            "first_call_seconds": times[0], "projected_per_s": st.projected / dt if dt > 0 else None}


def spawn_ranks(n):
    """Re-run this command under torch.distributed.run with n ranks on this node; fails loudly when the
    node has fewer than n GPUs (a silent 1-rank run would report a wrong n_gpus)."""
    import subprocess
    import impg_amd
    have = impg_amd.lib().impg_gpu_device_count()
    if have < n:
        print("bench.py: --gpus %d but this node has %d HIP device(s)" % (n, have), file=sys.stderr)
        return 2
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def cpu_baseline(args, paf, ranges, transitive):
    """The C++ restatement of the reference's CPU path (oracle, kind "port") timed on this box's host cores
    on bounded samples of the same workload, in the two modes of BASELINE.md section 3, each at T = all cores
    and T = 4 (the reference's default -t, main.rs:2031):
      (i)  reference-faithful: ranges serial (main.rs:7435), each BFS level's frontier parallel over a
           persistent worker pool (impg.rs:2384-2465), CIGARs re-read with pread + ASCII parse per hit
           (impg.rs:515-530, :2903-2950);
      (ii) in-memory: CIGARs pre-parsed to packed ops, ranges parallel.
    `value` is the best of the four; every leg is listed in `modes`."""
    from oracle import oracle as o
    cores = os.cpu_count() or 1
    p = o.make_params(transitive=transitive, max_depth=args.max_depth)
    legs = []
    budget_s = 5.0  # per leg

    def leg(ix, label, mode, threads):
        n0 = min(len(ranges), max(8, threads if mode == 1 else 8))
        proj, _, sec = ix.bench(ranges["target_id"][:n0], ranges["start"][:n0], ranges["end"][:n0], p, threads=threads, mode=mode)
        n = int(min(len(ranges), args.cpu_sample, max(n0, n0 * budget_s / max(sec, 1e-3))))
        if n > n0:
            proj, _, sec = ix.bench(ranges["target_id"][:n], ranges["start"][:n], ranges["end"][:n], p, threads=threads, mode=mode)
        else:
            n = n0
        legs.append({"mode": label, "threads": threads, "ranges": n, "projections": proj, "seconds": sec,
                     "value": proj / sec if sec > 0 else 0.0})
        log("cpu baseline: %s T=%d: %d ranges, %d projections in %.2f s" % (label, threads, n, proj, sec))

    t0 = time.time()
    ix = o.OracleIndex(paf_paths=[paf], preparse=False)
    build_s = time.time() - t0
    for T in sorted({cores, 4}, reverse=True):
        leg(ix, "reference-faithful (ranges serial, frontier parallel, per-hit pread + parse)", 0, T)
    del ix
    t0 = time.time()
    ix = o.OracleIndex(paf_paths=[paf], preparse=True)
    build2_s = time.time() - t0
    for T in sorted({cores, 4}, reverse=True):
        leg(ix, "in-memory (CIGARs pre-parsed, ranges parallel)", 1, T)
    best = max(legs, key=lambda l: l["value"])
    return {"value": best["value"], "unit": "projected ranges/s", "cores": best["threads"], "kind": "port",
            "sample": "first %d of the %d query ranges, same PAF and flags, %s (%d projections in %.2f s); C++ restatement of "
                      "the reference algorithm, not the impg binary" % (best["ranges"], len(ranges), best["mode"],
                                                                          best["projections"], best["seconds"]),
            "host_cores": cores, "modes": legs, "oracle_index_build_s": build_s, "oracle_preparse_build_s": build2_s}


if __name__ == "__main__":
    main()
