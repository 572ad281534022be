#!/usr/bin/env python3
"""A/B on the GPU box: the headline batch (100 000 ranges, -x -m 3) with the counting run's hit slots in
projection order (option free_slot_order = 1, the default) and in the reference's order (0); the per-range
counts and checksums must be identical."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 100_000)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
ref = None
for kw in [dict(transitive=True, max_depth=3), dict()]:
    p = impg_amd.make_params(**kw)
    ref = None
    # free_slot_order, regroup_entries (pairs regrouped by entry inside a projection block), fuse_final_level (the final
    # level's pairs enumerated from the count pass's windows: no emit pass)
    for free, regroup, fuse in [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 1, 0), (1, 1, 1)]:
        g.set_option("free_slot_order", free)
        g.set_option("regroup_entries", regroup)
        g.set_option("fuse_final_level", fuse)
        g.query_batch_stats(r, p)
        st, cnt, ck = g.query_batch_stats(r, p)
        sig = (cnt.tobytes(), ck.tobytes())
        if ref is None: ref = sig
        assert sig == ref
        print("%-40s free_slot_order %d regroup %d fuse %d: projected %d  lookup %.2f  project %.2f  update %.2f  total %.2f ms" %
              (kw, free, regroup, fuse, st.projected, st.ms_lookup, st.ms_project, st.ms_update, st.ms_total), flush=True)
