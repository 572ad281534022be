#!/usr/bin/env python3
"""The per-query walk kernel (walk_device.inc) on the headline index: per-call latency of BFS / DFS, and a DFS batch
in counting and in full-results form, each against the batch engine (option walk_kernel = 0) with identical counts."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
n_big = int(os.environ.get("WALK_BATCH", "100000"))
bed = impg_amd.synth_bed(7, max(4096, n_big))
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
cases = (("bfs -m 3", impg_amd.make_params(transitive=True, max_depth=3)), ("dfs -m 3", impg_amd.make_params(transitive=True, dfs=True, max_depth=3)),
         ("dfs -m 2", impg_amd.make_params(transitive=True, dfs=True, max_depth=2)))
for label, p in cases:
    for nb in (1, 8):
        out = []
        for walk in (1, 0):
            g.set_option("walk_kernel", walk)
            reps = 100 if walk else 10
            for _ in range(3):
                g.query_batch(ranges[:nb], p)
            t0 = time.perf_counter()
            rows = 0
            for k in range(reps):
                r = g.query_batch(ranges[(k * nb) % 4000:(k * nb) % 4000 + nb], p, copy=False)
                rows += r.total
            out.append(((time.perf_counter() - t0) / reps * 1e6, rows / reps))
        print("%-9s batch %2d: walk %9.1f us, batch engine %10.1f us per call (%6.0f / %6.0f rows)" % (label, nb, out[0][0], out[1][0], out[0][1], out[1][1]), flush=True)
g.set_option("walk_kernel", 1)
g.set_option("chunk_ranges", 50000)
g.set_option("pair_budget", 1 << 30)
for label, p in cases[1:]:
    for n in (1000, n_big):
        t0 = time.perf_counter()
        st, cnt, ck = g.query_batch_stats(ranges[:n], p)
        dt = time.perf_counter() - t0
        print("%-9s counting batch of %6d: %8.3f s, %d projections (%.3g / s)" % (label, n, dt, st.projected, st.projected / dt), flush=True)
        if n == 1000:
            g.set_option("walk_kernel", 0)
            t0 = time.perf_counter()
            st0, cnt0, ck0 = g.query_batch_stats(ranges[:n], p)
            print("          batch engine, same 1000: %8.3f s; identical counts %s checksums %s projections %s" %
                  (time.perf_counter() - t0, (cnt0 == cnt).all(), (ck0 == ck).all(), st0.projected == st.projected), flush=True)
            g.set_option("walk_kernel", 1)
            t0 = time.perf_counter()
            r = g.query_batch(ranges[:n], p, copy=False)
            print("          full results, same 1000: %8.3f s, %d rows (counts + self rows: %d)" % (time.perf_counter() - t0, r.total, int(cnt.sum()) + n), flush=True)
