#!/usr/bin/env python3
"""Per-call latency of the trait-shaped calls on the headline index: impg_gpu_query (one range, no transitive)
and query_transitive_bfs -m 3, plus small batches."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 4096)
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
for label, p in (("query", impg_amd.make_params()), ("bfs -m 3", impg_amd.make_params(transitive=True, max_depth=3)),
                 ("dfs -m 2", impg_amd.make_params(transitive=True, dfs=True, max_depth=2)),
                 # the shapes the per-query walk does not take (the batch engine answers them)
                 ("bfs3 cigar", impg_amd.make_params(transitive=True, max_depth=3, store_cigar=True)),
                 ("bfs3 multi", impg_amd.make_params(transitive=True, max_depth=3, multi_impg=True))):
    slow = label.startswith("bfs3 ")  # (answered by the batch engine: milliseconds to seconds a call -- a few calls only)
    for nb in ((1,) if slow else (1, 8, 64)):
        reps = 3 if slow else (200 if nb == 1 else 50)
        for _ in range(1 if slow else 5):
            g.query_batch(ranges[:nb], p)
        t0 = time.perf_counter()
        rows = 0
        for k in range(reps):
            r = g.query_batch(ranges[(k * nb) % 4000:(k * nb) % 4000 + nb], p, copy=False)
            rows += r.total
        dt = (time.perf_counter() - t0) / reps
        print("%-9s batch %2d: %8.1f us per call, %6.0f rows per call" % (label, nb, dt * 1e6, rows / reps))
