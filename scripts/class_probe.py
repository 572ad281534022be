#!/usr/bin/env python3
"""How much of the projection kernel is the two CIGAR walks?  Non-transitive batches whose ranges are much
shorter than the 10 kb alignments (nearly every pair needs both walks) vs much longer (most need none)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
g.set_option("pair_budget", 1 << 30)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
for rl, n in ((500, 8_000_000), (5000, 6_000_000), (40000, 2_000_000)):
    bed = impg_amd.synth_bed(7, n, range_len=rl)
    r = np.zeros(n, dtype=impg_amd.RANGE_DTYPE)
    r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
    p = impg_amd.make_params()
    g.query_batch_stats(r, p, counts=False, checksums=False)
    st, _, _ = g.query_batch_stats(r, p, counts=False, checksums=False)
    print("range_len %6d: pairs %d  lookup %.2f ms  project %.2f ms -> %.2f ns per 1000 pairs" %
          (rl, st.pairs, st.ms_lookup, st.ms_project, st.ms_project * 1e6 / st.pairs * 1e3 / 1e3))
