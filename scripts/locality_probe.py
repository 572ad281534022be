#!/usr/bin/env python3
"""Experiment: how much of lookup/project time is cache locality?  Same non-transitive
batch (identical work) with the ranges in random order vs sorted by (target, start)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
g.set_option("pair_budget", 1 << 29)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
bed = impg_amd.synth_bed(7, N)
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
ranges["target_id"] = ids[bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
order = np.lexsort((ranges["start"], ranges["target_id"]))
coarse = np.lexsort((ranges["start"] >> 14, ranges["target_id"]))  # stable: random within a 16 kb bin
p = impg_amd.make_params()
for name, r in [("random", ranges), ("sorted", ranges[order]), ("coarse-sorted(16kb bins)", ranges[coarse])]:
    r = np.ascontiguousarray(r)
    g.query_batch_stats(r, p, counts=False, checksums=False)
    st, _, _ = g.query_batch_stats(r, p, counts=False, checksums=False)
    print("%-26s pairs %d projected %d  lookup %.2f ms  project %.2f ms  total %.2f ms" %
          (name, st.pairs, st.projected, st.ms_lookup, st.ms_project, st.ms_total))
# frontier sizes of the headline chunk
bed = impg_amd.synth_bed(7, 16384)
r2 = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r2["target_id"] = ids[bed["target_id"]]; r2["start"], r2["end"] = bed["start"], bed["end"]
st, _, _ = g.query_batch_stats(r2, impg_amd.make_params(transitive=True, max_depth=3), counts=False, checksums=False)
print("headline chunk: frontier_ranges %d pairs %d projected %d levels %d lookup %.2f project %.2f update %.2f" %
      (st.frontier_ranges, st.pairs, st.projected, st.levels, st.ms_lookup, st.ms_project, st.ms_update))
