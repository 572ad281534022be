#!/usr/bin/env python3
"""DFS batches through the per-query walk kernel on the headline index (IMPG_WALK_DEBUG=1 prints the phase clocks)."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
sizes = [int(x) for x in os.environ.get("WALK_SIZES", "1,64,1000,4096").split(",")]
bed = impg_amd.synth_bed(7, max(sizes))
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
g.set_option("chunk_ranges", max(sizes))
for depth in (3,):
    p = impg_amd.make_params(transitive=True, dfs=True, max_depth=depth)
    for n in sizes:
        t0 = time.perf_counter()
        st, cnt, ck = g.query_batch_stats(ranges[:n], p)
        dt = time.perf_counter() - t0
        print("dfs -m %d n=%d: %.3f s, %d projections (%.3g / s)" % (depth, n, dt, st.projected, st.projected / dt), flush=True)
