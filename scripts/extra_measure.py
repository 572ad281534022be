#!/usr/bin/env python3
"""Side measurements for DESIGN.md (GPU box): BASELINE configs 2/3 (10k ranges),
per-call latency of the trait-shaped single-range query, PCIe-inclusive batch."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 10000)
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
for name, p in [("config2 no-transitive", impg_amd.make_params()),
                ("config3 -x -m 3", impg_amd.make_params(transitive=True, max_depth=3))]:
    g.query_batch_stats(ranges, p, counts=False, checksums=False)
    t = time.perf_counter()
    n = 5
    for _ in range(n):
        st, _, _ = g.query_batch_stats(ranges, p, counts=False, checksums=False)  # host ranges: includes the H2D copy
    dt = (time.perf_counter() - t) / n
    print("%s: %d projections in %.2f ms wall (engine %.2f ms) = %.3e /s, host ranges in" % (name, st.projected, dt * 1e3, st.ms_total, st.projected / dt))
# full results (D2H + host assembly) for config 2
t = time.perf_counter()
res = g.query_batch(ranges, impg_amd.make_params())
dt = time.perf_counter() - t
print("config2 full results to host: %d intervals in %.1f ms" % (len(res.intervals), dt * 1e3))
t = time.perf_counter()
txt = res.bed(None, merge_distance=1000, params=impg_amd.make_params())
print("config2 BED render (-d 1000): %d bytes in %.1f ms" % (len(txt), (time.perf_counter() - t) * 1e3))
# trait-shaped single-range calls
for name, kw in [("query", dict()), ("query_transitive_bfs -m 3", dict(transitive=True, max_depth=3))]:
    p = impg_amd.make_params(**kw)
    g.query_batch(ranges[:1], p)
    t = time.perf_counter()
    for i in range(50):
        g.query_batch(ranges[i:i + 1], p)
    print("single-range %s: %.3f ms per call" % (name, (time.perf_counter() - t) / 50 * 1e3))
