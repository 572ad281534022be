#!/usr/bin/env python3
"""A/B on the GPU box: headline chunk (16 384 ranges, -x -m 3) with and without the
locality order of the projection kernel; checks identical checksums."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
g.set_option("pair_budget", 1 << 29)
bed = impg_amd.synth_bed(7, 16384)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
p = impg_amd.make_params(transitive=True, max_depth=3)
ref = None
for loc in [0, 4096, 0, 4096]:
    g.set_option("locality_min", loc)
    g.query_batch_stats(r, p)
    st, cnt, ck = g.query_batch_stats(r, p)
    sig = (int(cnt.sum()), int(np.bitwise_xor.reduce(ck)))
    if ref is None: ref = sig
    assert sig == ref, (sig, ref)
    print("locality_min %5d: projected %d  lookup %.2f  project %.2f  update %.2f  total %.2f ms" %
          (loc, st.projected, st.ms_lookup, st.ms_project, st.ms_update, st.ms_total))
